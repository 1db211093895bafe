/*
 * jpeg_oracle.c -- TEST INFRASTRUCTURE ONLY.  A plain-C, single-threaded CPU restatement of the
 * arithmetic that the reference's CPU JPEG path produces, used as the bit-exact checker for
 * the HIP kernels.  Nothing under nvimagecodec_amd/ may include, link or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * What it restates.  The reference decodes JPEG on the CPU in
 *   extensions/libjpeg_turbo/libjpeg_turbo_decoder.cpp:324-442  (format mapping, stride handling)
 *   extensions/libjpeg_turbo/jpeg_mem.cpp:102-460               (libjpeg session: JCS_RGB / JCS_EXT_BGR /
 *                                                                JCS_GRAYSCALE, JDCT_ISLOW, do_fancy_upsampling)
 * and every arithmetic step is delegated to libjpeg-turbo (pinned 3.0.1, external/README.rst:98-104),
 * an un-vendored submodule that is NOT under /root/reference.  The steps below therefore restate
 * libjpeg-turbo's published algorithm (file names of that library given per function):
 *   entropy decode    jdhuff.c / jdphuff.c   (ITU T.81 Annex F / G)
 *   dequant + IDCT    jpeg_idct_islow (CONST_BITS 13, PASS1_BITS 2) as the x86-64 SIMD builds run it (jidctint-sse2/avx2.asm; default)
 *                     and as jidctint.c + the range-limit table of jdmaster.c do (oj_set_idct_variant(1)); equal on every stream an
 *                     encoder can write, different out of gamut -- see oj_idct_block_simd
 *   upsampling        jdsample.c h2v1_fancy / h2v2_fancy / h1v2_fancy / int_upsample (replication)
 *   colour            jdcolor.c ycc_rgb_convert (SCALEBITS 16), gray->rgb, rgb passthrough
 *   encode            jccolor.c, jcsample.c, jfdctint.c, jcdctmgr.c, jcparam.c, jchuff.c, jcmarker.c
 *
 * Pinning.  tests/test_oracle_golden.py checks this file against golden vectors produced by the real
 * libjpeg-turbo (3.1.4.1 bundled in Pillow 12.2.0; generator tests/golden/make_golden.py).  The reference's
 * own fixtures for this path (resources/ref/jpeg/ *.ppm) cannot be replayed: their input .jpg files are
 * git-LFS stubs in this snapshot.
 *
 * Scope: 8-bit Huffman JPEG, SOF0/SOF1/SOF2, 1 or 3 components (gray / YCbCr / Adobe-RGB), restart
 * intervals, multi-scan, four-component CMYK/YCCK frames.  Arithmetic coding, 12-bit, lossless are rejected (the real
 * framework hands those to another decoder).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OJ_OK 0
#define OJ_ERR_SYNTAX -1
#define OJ_ERR_UNSUPPORTED -2
#define OJ_ERR_TRUNCATED -3
#define OJ_ERR_ARG -4

#define OJ_CS_GRAY 0
#define OJ_CS_YCC 1
#define OJ_CS_RGB 2
#define OJ_CS_CMYK 3
#define OJ_CS_YCCK 4

/* output formats of oj_decode */
#define OJ_FMT_RGB 0  /* interleaved R,G,B */
#define OJ_FMT_BGR 1  /* interleaved B,G,R (JCS_EXT_BGR) */
#define OJ_FMT_GRAY 2 /* single plane (JCS_GRAYSCALE) */

static const uint8_t oj_zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                      41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                      15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef struct {
    uint8_t bits[17];
    uint8_t vals[256];
    int present;
    /* derived (T.81 F.2.2.3) */
    int32_t maxcode[18];
    int32_t valoff[17];
    uint8_t look_sym[256];
    uint8_t look_len[256]; /* 0 = needs slow path */
} oj_huff;

typedef struct {
    int id, h, v, tq;
    int td, ta;
    int bw, bh; /* allocated size in blocks (padded to whole MCUs) */
    int dw, dh; /* true component size in samples: ceil(W*h/hmax), ceil(H*v/vmax) */
    int16_t* coef; /* [bh][bw][64], natural (row-major) order inside a block */
} oj_comp;

typedef struct {
    int width, height, ncomp, precision;
    int sof; /* 0xC0, 0xC1, 0xC2 */
    int hmax, vmax, mcux, mcuy;
    int restart_interval;
    int saw_jfif, saw_adobe, adobe_transform;
    int colorspace;
    uint16_t qt[4][64]; /* natural order */
    int qt_present[4];
    oj_huff dc[4], ac[4];
    oj_comp comp[4];
    /* per-component snapshot of the quant table at the time of its first scan (jdinput.c latch_quant_tables) */
    uint16_t cq[4][64];
    int cq_latched[4];
} oj_dec;

/* ---------------------------------------------------------------- public info struct */
typedef struct {
    int32_t width, height, ncomp, sof, colorspace, restart_interval;
    int32_t h[4], v[4];
    int32_t bw[4], bh[4], dw[4], dh[4];
    int32_t hmax, vmax;
} oj_info;

/* ---------------------------------------------------------------- Huffman tables */
static int oj_build_huff(oj_huff* t)
{
    int code = 0, k = 0, i, l;
    for (l = 1; l <= 16; l++) {
        t->valoff[l] = k - code;
        if (t->bits[l]) {
            k += t->bits[l];
            code += t->bits[l];
            t->maxcode[l] = code - 1;
        } else {
            t->maxcode[l] = -1;
        }
        if (code >= (1 << l) && code > 0) return OJ_ERR_SYNTAX; /* jdhuff.c: "no code is allowed to be all ones" */
        code <<= 1;
    }
    t->maxcode[17] = 0x7fffffff;
    if (k > 256) return OJ_ERR_SYNTAX;
    memset(t->look_len, 0, sizeof t->look_len);
    code = 0;
    k = 0;
    for (l = 1; l <= 8; l++) {
        for (i = 0; i < t->bits[l]; i++, k++, code++) {
            int lo = code << (8 - l), n = 1 << (8 - l), j;
            for (j = 0; j < n; j++) {
                t->look_sym[lo + j] = t->vals[k];
                t->look_len[lo + j] = (uint8_t)l;
            }
        }
        code <<= 1;
    }
    t->present = 1;
    return OJ_OK;
}

/* ---------------------------------------------------------------- bit reader (jdhuff.c fill_bit_buffer) */
typedef struct {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t buf;
    int cnt;
    int hit_marker; /* marker byte seen in entropy data (0 = none) */
    int pad_bits;   /* zero bits appended after the data ran out (at the tail of buf) */
} oj_bits;

static void oj_fill(oj_bits* b)
{
    while (b->cnt <= 56) {
        int c = 0;
        if (!b->hit_marker && b->p < b->end) {
            c = *b->p++;
            if (c == 0xFF) {
                int c2;
                /* skip fill FFs */
                while (b->p < b->end && *b->p == 0xFF) b->p++;
                if (b->p >= b->end) {
                    b->hit_marker = 0xD9;
                    c = 0;
                } else {
                    c2 = *b->p;
                    if (c2 == 0) {
                        b->p++;
                        c = 0xFF;
                    } else {
                        b->hit_marker = c2; /* leave p on the marker code */
                        c = 0;
                    }
                }
            }
        } else {
            if (!b->hit_marker) b->hit_marker = 0xD9; /* ran off the end: behave like EOI */
            b->pad_bits += 8;
        }
        b->buf |= (uint64_t)c << (56 - b->cnt);
        b->cnt += 8;
    }
}

static inline int oj_getbits(oj_bits* b, int n)
{
    int v;
    if (n == 0) return 0;
    if (b->cnt < n) oj_fill(b);
    v = (int)(b->buf >> (64 - n));
    b->buf <<= n;
    b->cnt -= n;
    return v;
}

static inline int oj_getbit(oj_bits* b) { return oj_getbits(b, 1); }

static int oj_decode_sym(oj_bits* b, const oj_huff* t)
{
    int look, l, code;
    if (b->cnt < 16) oj_fill(b);
    look = (int)(b->buf >> 56);
    l = t->look_len[look];
    if (l) {
        b->buf <<= l;
        b->cnt -= l;
        return t->look_sym[look];
    }
    code = (int)(b->buf >> (64 - 9));
    for (l = 9; l <= 16; l++) {
        if (code <= t->maxcode[l] && t->maxcode[l] >= 0) break;
        code = (int)(b->buf >> (64 - (l + 1)));
    }
    if (l > 16) return -1;
    b->buf <<= l;
    b->cnt -= l;
    return t->vals[(code + t->valoff[l]) & 0xFF];
}

/* T.81 F.2.2.1 EXTEND */
static inline int oj_extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

/* ---------------------------------------------------------------- marker parsing (jdmarker.c) */
static int oj_u16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

static void oj_free(oj_dec* d)
{
    int c;
    for (c = 0; c < 4; c++) {
        free(d->comp[c].coef);
        d->comp[c].coef = NULL;
    }
}

static int oj_setup_frame(oj_dec* d)
{
    int c;
    d->hmax = d->vmax = 1;
    for (c = 0; c < d->ncomp; c++) {
        if (d->comp[c].h < 1 || d->comp[c].h > 4 || d->comp[c].v < 1 || d->comp[c].v > 4) return OJ_ERR_SYNTAX;
        if (d->comp[c].h > d->hmax) d->hmax = d->comp[c].h;
        if (d->comp[c].v > d->vmax) d->vmax = d->comp[c].v;
    }
    d->mcux = (d->width + 8 * d->hmax - 1) / (8 * d->hmax);
    d->mcuy = (d->height + 8 * d->vmax - 1) / (8 * d->vmax);
    for (c = 0; c < d->ncomp; c++) {
        oj_comp* k = &d->comp[c];
        k->bw = d->mcux * k->h;
        k->bh = d->mcuy * k->v;
        k->dw = (d->width * k->h + d->hmax - 1) / d->hmax;
        k->dh = (d->height * k->v + d->vmax - 1) / d->vmax;
    }
    /* colour space guess: jdapimin.c default_decompress_parms */
    if (d->ncomp == 1) {
        d->colorspace = OJ_CS_GRAY;
    } else if (d->ncomp == 3) {
        if (d->saw_jfif) {
            d->colorspace = OJ_CS_YCC;
        } else if (d->saw_adobe) {
            d->colorspace = d->adobe_transform == 0 ? OJ_CS_RGB : OJ_CS_YCC;
        } else {
            int a = d->comp[0].id, b = d->comp[1].id, e = d->comp[2].id;
            if (a == 1 && b == 2 && e == 3)
                d->colorspace = OJ_CS_YCC;
            else if (a == 82 && b == 71 && e == 66)
                d->colorspace = OJ_CS_RGB;
            else
                d->colorspace = OJ_CS_YCC;
        }
    } else if (d->ncomp == 4) {
        d->colorspace = (d->saw_adobe && d->adobe_transform == 2) ? OJ_CS_YCCK : OJ_CS_CMYK;
    } else {
        return OJ_ERR_UNSUPPORTED;
    }
    return OJ_OK;
}

/* ---------------------------------------------------------------- scans */
typedef struct {
    int ncomp;
    int ci[4];
    int ss, se, ah, al;
} oj_scan;

static int oj_restart(oj_bits* b, int* next_rst)
{
    /* byte align, expect RSTn (jdhuff.c process_restart).  Bits left in the buffer are discarded. */
    if (b->cnt < b->pad_bits) return OJ_ERR_TRUNCATED; /* consumed invented bits */
    b->buf = 0;
    b->cnt = 0;
    b->pad_bits = 0;
    if (!b->hit_marker) {
        /* scan forward to the marker */
        while (b->p + 1 < b->end && !(b->p[0] == 0xFF && b->p[1] != 0 && b->p[1] != 0xFF)) b->p++;
        if (b->p + 1 >= b->end) return OJ_ERR_TRUNCATED;
        b->p++;
        b->hit_marker = *b->p;
    }
    if (b->hit_marker != 0xD0 + *next_rst) return OJ_ERR_SYNTAX;
    b->p++; /* consume marker code */
    b->hit_marker = 0;
    *next_rst = (*next_rst + 1) & 7;
    return OJ_OK;
}

static int oj_decode_block_seq(oj_bits* b, const oj_huff* dc, const oj_huff* ac, int* pred, int16_t* blk)
{
    int s, k, r;
    s = oj_decode_sym(b, dc);
    if (s < 0 || s > 15) return OJ_ERR_SYNTAX;
    if (s) {
        r = oj_getbits(b, s);
        s = oj_extend(r, s);
    }
    *pred += s;
    blk[0] = (int16_t)*pred;
    for (k = 1; k < 64;) {
        int rs = oj_decode_sym(b, ac);
        if (rs < 0) return OJ_ERR_SYNTAX;
        r = rs >> 4;
        s = rs & 15;
        if (s) {
            k += r;
            if (k > 63) return OJ_ERR_SYNTAX;
            r = oj_getbits(b, s);
            blk[oj_zigzag[k]] = (int16_t)oj_extend(r, s);
            k++;
        } else {
            if (r != 15) break;
            k += 16;
        }
    }
    return OJ_OK;
}

/* jdphuff.c decode_mcu_AC_first */
static int oj_ac_first(oj_bits* b, const oj_huff* ac, int16_t* blk, int ss, int se, int al, unsigned* eobrun)
{
    int k, r, s;
    if (*eobrun > 0) {
        (*eobrun)--;
        return OJ_OK;
    }
    for (k = ss; k <= se; k++) {
        int rs = oj_decode_sym(b, ac);
        if (rs < 0) return OJ_ERR_SYNTAX;
        r = rs >> 4;
        s = rs & 15;
        if (s) {
            k += r;
            if (k > 63) return OJ_ERR_SYNTAX;
            r = oj_getbits(b, s);
            blk[oj_zigzag[k]] = (int16_t)(oj_extend(r, s) * (1 << al));
        } else {
            if (r == 15) {
                k += 15;
            } else {
                *eobrun = 1u << r;
                if (r) *eobrun += (unsigned)oj_getbits(b, r);
                (*eobrun)--;
                break;
            }
        }
    }
    return OJ_OK;
}

/* jdphuff.c decode_mcu_AC_refine */
static int oj_ac_refine(oj_bits* b, const oj_huff* ac, int16_t* blk, int ss, int se, int al, unsigned* eobrun)
{
    int p1 = 1 << al, m1 = -(1 << al);
    int k = ss, r, s;
    if (*eobrun == 0) {
        for (; k <= se; k++) {
            int rs = oj_decode_sym(b, ac);
            if (rs < 0) return OJ_ERR_SYNTAX;
            r = rs >> 4;
            s = rs & 15;
            if (s) {
                if (s != 1) return OJ_ERR_SYNTAX;
                s = oj_getbit(b) ? p1 : m1;
            } else if (r != 15) {
                *eobrun = 1u << r;
                if (r) *eobrun += (unsigned)oj_getbits(b, r);
                break;
            }
            /* advance over already-nonzero coefs and r still-zero coefs, appending correction bits */
            do {
                int16_t* c = &blk[oj_zigzag[k]];
                if (*c != 0) {
                    if (oj_getbit(b)) {
                        if ((*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1);
                    }
                } else {
                    if (--r < 0) break;
                }
                k++;
            } while (k <= se);
            if (s) {
                if (k > 63) return OJ_ERR_SYNTAX;
                blk[oj_zigzag[k]] = (int16_t)s;
            }
        }
    }
    if (*eobrun > 0) {
        for (; k <= se; k++) {
            int16_t* c = &blk[oj_zigzag[k]];
            if (*c != 0) {
                if (oj_getbit(b)) {
                    if ((*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1);
                }
            }
        }
        (*eobrun)--;
    }
    return OJ_OK;
}

static int oj_decode_scan(oj_dec* d, const oj_scan* sc, const uint8_t* p, const uint8_t* end, const uint8_t** next)
{
    oj_bits b;
    int pred[4] = {0, 0, 0, 0};
    unsigned eobrun = 0;
    int progressive = d->sof == 0xC2;
    int rst_left = d->restart_interval, next_rst = 0;
    int mx, my, i, rc;
    memset(&b, 0, sizeof b);
    b.p = p;
    b.end = end;

    for (i = 0; i < sc->ncomp; i++) {
        oj_comp* k = &d->comp[sc->ci[i]];
        int need_dc = !progressive || sc->ss == 0;
        int need_ac = !progressive || sc->ss > 0;
        if (need_dc && !(progressive && sc->ah) && !d->dc[k->td].present) return OJ_ERR_SYNTAX;
        if (need_ac && !d->ac[k->ta].present) return OJ_ERR_SYNTAX;
        if (!d->cq_latched[sc->ci[i]]) {
            if (!d->qt_present[k->tq]) return OJ_ERR_SYNTAX;
            memcpy(d->cq[sc->ci[i]], d->qt[k->tq], sizeof d->cq[0]);
            d->cq_latched[sc->ci[i]] = 1;
        }
    }

    if (sc->ncomp == 1) {
        /* non-interleaved: one block per MCU, only the blocks that cover real samples (T.81 A.2.2) */
        oj_comp* k = &d->comp[sc->ci[0]];
        int nbx = (k->dw + 7) / 8, nby = (k->dh + 7) / 8;
        for (my = 0; my < nby; my++)
            for (mx = 0; mx < nbx; mx++) {
                int16_t* blk = k->coef + ((size_t)my * k->bw + mx) * 64;
                if (d->restart_interval && rst_left == 0) {
                    if ((rc = oj_restart(&b, &next_rst)) != OJ_OK) return rc;
                    pred[0] = 0;
                    eobrun = 0;
                    rst_left = d->restart_interval;
                }
                if (!progressive) {
                    rc = oj_decode_block_seq(&b, &d->dc[k->td], &d->ac[k->ta], &pred[0], blk);
                } else if (sc->ss == 0) {
                    if (sc->ah == 0) {
                        int s = oj_decode_sym(&b, &d->dc[k->td]);
                        if (s < 0 || s > 15) return OJ_ERR_SYNTAX;
                        if (s) s = oj_extend(oj_getbits(&b, s), s);
                        pred[0] += s;
                        blk[0] = (int16_t)(pred[0] * (1 << sc->al));
                    } else if (oj_getbit(&b)) {
                        blk[0] |= (int16_t)(1 << sc->al);
                    }
                    rc = OJ_OK;
                } else if (sc->ah == 0) {
                    rc = oj_ac_first(&b, &d->ac[k->ta], blk, sc->ss, sc->se, sc->al, &eobrun);
                } else {
                    rc = oj_ac_refine(&b, &d->ac[k->ta], blk, sc->ss, sc->se, sc->al, &eobrun);
                }
                if (rc != OJ_OK) return rc;
                rst_left--;
            }
    } else {
        if (progressive && sc->ss != 0) return OJ_ERR_SYNTAX; /* AC scans are never interleaved */
        for (my = 0; my < d->mcuy; my++)
            for (mx = 0; mx < d->mcux; mx++) {
                if (d->restart_interval && rst_left == 0) {
                    if ((rc = oj_restart(&b, &next_rst)) != OJ_OK) return rc;
                    memset(pred, 0, sizeof pred);
                    rst_left = d->restart_interval;
                }
                for (i = 0; i < sc->ncomp; i++) {
                    oj_comp* k = &d->comp[sc->ci[i]];
                    int bx, by;
                    for (by = 0; by < k->v; by++)
                        for (bx = 0; bx < k->h; bx++) {
                            int16_t* blk = k->coef + ((size_t)(my * k->v + by) * k->bw + (mx * k->h + bx)) * 64;
                            if (!progressive) {
                                rc = oj_decode_block_seq(&b, &d->dc[k->td], &d->ac[k->ta], &pred[i], blk);
                                if (rc != OJ_OK) return rc;
                            } else if (sc->ah == 0) {
                                int s = oj_decode_sym(&b, &d->dc[k->td]);
                                if (s < 0 || s > 15) return OJ_ERR_SYNTAX;
                                if (s) s = oj_extend(oj_getbits(&b, s), s);
                                pred[i] += s;
                                blk[0] = (int16_t)(pred[i] * (1 << sc->al));
                            } else if (oj_getbit(&b)) {
                                blk[0] |= (int16_t)(1 << sc->al);
                            }
                        }
                }
                rst_left--;
            }
    }
    if (b.cnt < b.pad_bits) return OJ_ERR_TRUNCATED; /* the scan consumed bits that were not in the file */
    /* position of the next marker: bytes not yet pulled into the bit buffer are never markers we skipped */
    if (b.hit_marker)
        *next = b.p - 1; /* p sits on the marker code; back up to its 0xFF */
    else {
        const uint8_t* q = b.p;
        while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
        *next = q;
    }
    return OJ_OK;
}

/* Parse all markers; decode entropy data if want_coef. */
static int oj_parse(oj_dec* d, const uint8_t* data, size_t len, int want_coef)
{
    const uint8_t* p = data;
    const uint8_t* end = data + len;
    int got_sof = 0, scans = 0, c, rc;
    memset(d, 0, sizeof *d);
    d->adobe_transform = -1;
    if (len < 4 || p[0] != 0xFF || p[1] != 0xD8) return OJ_ERR_SYNTAX;
    p += 2;
    for (;;) {
        int m, L;
        const uint8_t* seg;
        /* next marker */
        while (p < end && *p != 0xFF) p++;
        while (p < end && *p == 0xFF) p++;
        if (p >= end) break;
        m = *p++;
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (p + 2 > end) return OJ_ERR_TRUNCATED;
        L = oj_u16(p);
        if (L < 2 || p + L > end) return OJ_ERR_TRUNCATED;
        seg = p + 2;
        switch (m) {
        case 0xE0:
            if (L >= 7 && !memcmp(seg, "JFIF", 5)) d->saw_jfif = 1;
            break;
        case 0xEE:
            if (L >= 14 && !memcmp(seg, "Adobe", 5)) {
                d->saw_adobe = 1;
                d->adobe_transform = seg[11];
            }
            break;
        case 0xDB: {
            const uint8_t* q = seg;
            const uint8_t* qe = p + L;
            while (q < qe) {
                int pq = *q >> 4, tq = *q & 15, i;
                q++;
                if (tq > 3 || pq > 1) return OJ_ERR_SYNTAX;
                if (q + 64 * (pq + 1) > qe) return OJ_ERR_SYNTAX;
                for (i = 0; i < 64; i++) {
                    int v = pq ? oj_u16(q + 2 * i) : q[i];
                    d->qt[tq][oj_zigzag[i]] = (uint16_t)v;
                }
                d->qt_present[tq] = 1;
                q += 64 * (pq + 1);
            }
            break;
        }
        case 0xC4: {
            const uint8_t* q = seg;
            const uint8_t* qe = p + L;
            while (q < qe) {
                int tc = *q >> 4, th = *q & 15, i, n = 0;
                oj_huff* t;
                q++;
                if (tc > 1 || th > 3 || q + 16 > qe) return OJ_ERR_SYNTAX;
                t = tc ? &d->ac[th] : &d->dc[th];
                memset(t, 0, sizeof *t);
                for (i = 1; i <= 16; i++) {
                    t->bits[i] = q[i - 1];
                    n += q[i - 1];
                }
                q += 16;
                if (n > 256 || q + n > qe) return OJ_ERR_SYNTAX;
                memcpy(t->vals, q, (size_t)n);
                q += n;
                if ((rc = oj_build_huff(t)) != OJ_OK) return rc;
            }
            break;
        }
        case 0xDD:
            if (L != 4) return OJ_ERR_SYNTAX;
            d->restart_interval = oj_u16(seg);
            break;
        case 0xC0:
        case 0xC1:
        case 0xC2: {
            if (got_sof) return OJ_ERR_SYNTAX;
            if (L < 8) return OJ_ERR_SYNTAX;
            d->sof = m;
            d->precision = seg[0];
            d->height = oj_u16(seg + 1);
            d->width = oj_u16(seg + 3);
            d->ncomp = seg[5];
            if (d->precision != 8) return OJ_ERR_UNSUPPORTED;
            if (d->width == 0 || d->height == 0) return OJ_ERR_UNSUPPORTED;
            if (d->ncomp < 1 || d->ncomp > 4 || L != 8 + 3 * d->ncomp) return OJ_ERR_SYNTAX;
            for (c = 0; c < d->ncomp; c++) {
                d->comp[c].id = seg[6 + 3 * c];
                d->comp[c].h = seg[7 + 3 * c] >> 4;
                d->comp[c].v = seg[7 + 3 * c] & 15;
                d->comp[c].tq = seg[8 + 3 * c];
                if (d->comp[c].tq > 3) return OJ_ERR_SYNTAX;
            }
            if ((rc = oj_setup_frame(d)) != OJ_OK) return rc;
            got_sof = 1;
            break;
        }
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            return OJ_ERR_UNSUPPORTED;
        case 0xDA: {
            oj_scan sc;
            const uint8_t* nx;
            int i;
            if (!got_sof) return OJ_ERR_SYNTAX;
            if (!want_coef) return OJ_OK;
            sc.ncomp = seg[0];
            if (sc.ncomp < 1 || sc.ncomp > 4 || L != 6 + 2 * sc.ncomp) return OJ_ERR_SYNTAX;
            for (i = 0; i < sc.ncomp; i++) {
                int cid = seg[1 + 2 * i], found = -1;
                for (c = 0; c < d->ncomp; c++)
                    if (d->comp[c].id == cid) found = c;
                if (found < 0) return OJ_ERR_SYNTAX;
                sc.ci[i] = found;
                d->comp[found].td = seg[2 + 2 * i] >> 4;
                d->comp[found].ta = seg[2 + 2 * i] & 15;
                if (d->comp[found].td > 3 || d->comp[found].ta > 3) return OJ_ERR_SYNTAX;
            }
            sc.ss = seg[1 + 2 * sc.ncomp];
            sc.se = seg[2 + 2 * sc.ncomp];
            sc.ah = seg[3 + 2 * sc.ncomp] >> 4;
            sc.al = seg[3 + 2 * sc.ncomp] & 15;
            if (d->sof != 0xC2) {
                sc.ss = 0;
                sc.se = 63;
                sc.ah = sc.al = 0;
            } else if (sc.ss > sc.se || sc.se > 63 || sc.al > 13) {
                return OJ_ERR_SYNTAX;
            }
            if (!scans) {
                for (c = 0; c < d->ncomp; c++) {
                    oj_comp* k = &d->comp[c];
                    k->coef = (int16_t*)calloc((size_t)k->bw * k->bh * 64, sizeof(int16_t));
                    if (!k->coef) return OJ_ERR_ARG;
                }
            }
            scans++;
            rc = oj_decode_scan(d, &sc, p + L, end, &nx);
            if (rc != OJ_OK) return rc;
            p = nx;
            continue;
        }
        default:
            break;
        }
        p += L;
    }
    if (!got_sof) return OJ_ERR_SYNTAX;
    if (want_coef && !scans) return OJ_ERR_SYNTAX;
    return OJ_OK;
}

/* ---------------------------------------------------------------- IDCT (jidctint.c jpeg_idct_islow) */
#define F_0_298 2446
#define F_0_390 3196
#define F_0_541 4433
#define F_0_765 6270
#define F_0_899 7373
#define F_1_175 9633
#define F_1_501 12299
#define F_1_847 15137
#define F_1_961 16069
#define F_2_053 16819
#define F_2_562 20995
#define F_3_072 25172

#define OJ_DESCALE(x, n) (((x) + ((int32_t)1 << ((n)-1))) >> (n))

/* jidctint.c computes in JLONG -- `long`, 64 bits on the LP64 targets the reference builds for -- and stores pass 1 into an `int`
 * workspace; nothing wraps before that store. */
static void oj_idct1d(const int32_t in[8], int32_t out[8], int shift)
{
    int64_t z1, z2, z3, z4, z5, tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13;
    const int64_t rnd = (int64_t)1 << (shift - 1);
    z2 = in[2];
    z3 = in[6];
    z1 = (z2 + z3) * F_0_541;
    tmp2 = z1 + z3 * -F_1_847;
    tmp3 = z1 + z2 * F_0_765;
    tmp0 = ((int64_t)in[0] + in[4]) * 8192;
    tmp1 = ((int64_t)in[0] - in[4]) * 8192;
    tmp10 = tmp0 + tmp3;
    tmp13 = tmp0 - tmp3;
    tmp11 = tmp1 + tmp2;
    tmp12 = tmp1 - tmp2;
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    z4 = tmp1 + tmp3;
    z5 = (z3 + z4) * F_1_175;
    tmp0 *= F_0_298;
    tmp1 *= F_2_053;
    tmp2 *= F_3_072;
    tmp3 *= F_1_501;
    z1 *= -F_0_899;
    z2 *= -F_2_562;
    z3 = z3 * -F_1_961 + z5;
    z4 = z4 * -F_0_390 + z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    out[0] = (int32_t)((tmp10 + tmp3 + rnd) >> shift);
    out[7] = (int32_t)((tmp10 - tmp3 + rnd) >> shift);
    out[1] = (int32_t)((tmp11 + tmp2 + rnd) >> shift);
    out[6] = (int32_t)((tmp11 - tmp2 + rnd) >> shift);
    out[2] = (int32_t)((tmp12 + tmp1 + rnd) >> shift);
    out[5] = (int32_t)((tmp12 - tmp1 + rnd) >> shift);
    out[3] = (int32_t)((tmp13 + tmp0 + rnd) >> shift);
    out[4] = (int32_t)((tmp13 - tmp0 + rnd) >> shift);
}

/* post-IDCT range-limit table lookup (jdmaster.c prepare_range_limit_table), index = v & 1023 */
static inline uint8_t oj_range_limit(int32_t v)
{
    int i = v & 1023;
    if (i < 128) return (uint8_t)(i + 128);
    if (i < 512) return 255;
    if (i < 896) return 0;
    return (uint8_t)(i - 896);
}

static void oj_idct_block_c(const int16_t* coef, const uint16_t* q, uint8_t* out, int out_stride)
{
    int32_t ws[64], in[8], o[8];
    int c, r;
    for (c = 0; c < 8; c++) {
        /* jidctint.c DEQUANTIZE: ((ISLOW_MULT_TYPE)coef) * quantval with both operands 16-bit (jddctmgr.c stores the table as short) */
        for (r = 0; r < 8; r++) in[r] = (int32_t)coef[r * 8 + c] * (int32_t)(int16_t)q[r * 8 + c];
        oj_idct1d(in, o, 11); /* CONST_BITS - PASS1_BITS */
        for (r = 0; r < 8; r++) ws[r * 8 + c] = o[r];
    }
    for (r = 0; r < 8; r++) {
        oj_idct1d(&ws[r * 8], o, 18); /* CONST_BITS + PASS1_BITS + 3 */
        for (c = 0; c < 8; c++) out[r * out_stride + c] = oj_range_limit(o[c]);
    }
}

/* ---------------------------------------------------------------- IDCT as libjpeg-turbo's x86-64 SIMD builds compute it
 * (simd/x86_64/jidctint-sse2.asm / jidctint-avx2.asm, jsimd_idct_islow_*): the function the reference's CPU path actually runs --
 * extensions/libjpeg_turbo/jpeg_mem.cpp:174-177 selects JDCT_ISLOW and external/build_libjpeg-turbo.sh:36-39 builds the library with
 * its defaults (WITH_SIMD on).  For every stream an encoder can produce it equals jidctint.c; out of gamut it does not, because it
 * works on 16-bit lanes:
 *   - dequantization is pmullw: the low 16 bits of coefficient x quantizer;
 *   - if rows 1..7 of the (undequantized) block are all zero, pass 1 is skipped: workspace = row 0 << PASS1_BITS in 16 bits (psllw, wraps);
 *   - in0 +- in4 and the odd part's z3 = in7 + in3, z4 = in5 + in1 are 16-bit additions (paddw / psubw, wrap);
 *   - all products are pmaddwd on (value, value) x (constant, constant) pairs, exact in 32 bits, sums never leave int32;
 *   - pass 1 results are packed with signed saturation to int16 (packssdw), pass 2 results saturate to int16, then to int8
 *     (packsswb) and get +128 (paddb): a true clamp instead of jidctint.c's range-limit table indexed modulo 1024.
 * Restated from the published algorithm and pinned by vectors from the real library (tests/golden/make_golden_simd_idct.py: Pillow's
 * libjpeg-turbo 3.1.4.1 in its default AVX2 dispatch and with JSIMD_FORCESSE2=1 -- identical pictures; JSIMD_FORCENONE=1 gives
 * oj_idct_block_c's). */
static inline int32_t w16(int32_t v) { return (int32_t)(int16_t)(uint16_t)v; }
static inline int32_t s16(int32_t v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

static void oj_idct1d_simd(const int32_t in[8], int32_t out[8], int shift)
{
    /* inputs are int16 values; every sum below fits int32 (|.| < 1.7e9) */
    const int32_t rnd = (int32_t)1 << (shift - 1);
    int32_t tmp3 = in[2] * (F_0_541 + F_0_765) + in[6] * F_0_541;
    int32_t tmp2 = in[2] * F_0_541 + in[6] * (F_0_541 - F_1_847);
    int32_t tmp0 = w16(in[0] + in[4]) * 8192;
    int32_t tmp1 = w16(in[0] - in[4]) * 8192;
    int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int32_t z3 = w16(in[7] + in[3]), z4 = w16(in[5] + in[1]);
    int32_t z3n = z3 * (F_1_175 - F_1_961) + z4 * F_1_175;
    int32_t z4n = z3 * F_1_175 + z4 * (F_1_175 - F_0_390);
    int32_t t0 = in[7] * (F_0_298 - F_0_899) + in[1] * -F_0_899 + z3n;
    int32_t t3 = in[7] * -F_0_899 + in[1] * (F_1_501 - F_0_899) + z4n;
    int32_t t1 = in[5] * (F_2_053 - F_2_562) + in[3] * -F_2_562 + z4n;
    int32_t t2 = in[5] * -F_2_562 + in[3] * (F_3_072 - F_2_562) + z3n;
    out[0] = s16((tmp10 + t3 + rnd) >> shift);
    out[7] = s16((tmp10 - t3 + rnd) >> shift);
    out[1] = s16((tmp11 + t2 + rnd) >> shift);
    out[6] = s16((tmp11 - t2 + rnd) >> shift);
    out[2] = s16((tmp12 + t1 + rnd) >> shift);
    out[5] = s16((tmp12 - t1 + rnd) >> shift);
    out[3] = s16((tmp13 + t0 + rnd) >> shift);
    out[4] = s16((tmp13 - t0 + rnd) >> shift);
}

static void oj_idct_block_simd(const int16_t* coef, const uint16_t* q, uint8_t* out, int out_stride)
{
    int32_t ws[64], in[8], o[8];
    int c, r, k, ac_rows = 0;
    for (k = 8; k < 64; k++) ac_rows |= coef[k];
    if (!ac_rows) {
        for (c = 0; c < 8; c++) {
            int32_t v = w16(w16((int32_t)coef[c] * (int32_t)q[c]) * 4);
            for (r = 0; r < 8; r++) ws[r * 8 + c] = v;
        }
    } else {
        for (c = 0; c < 8; c++) {
            for (r = 0; r < 8; r++) in[r] = w16((int32_t)coef[r * 8 + c] * (int32_t)q[r * 8 + c]);
            oj_idct1d_simd(in, o, 11);
            for (r = 0; r < 8; r++) ws[r * 8 + c] = o[r];
        }
    }
    for (r = 0; r < 8; r++) {
        oj_idct1d_simd(&ws[r * 8], o, 18);
        for (c = 0; c < 8; c++) {
            int32_t v = o[c] < -128 ? -128 : (o[c] > 127 ? 127 : o[c]);
            out[r * out_stride + c] = (uint8_t)(v + 128);
        }
    }
}

/* 0 = the SIMD builds' arithmetic (default: what the reference's libjpeg-turbo runs on x86-64), 1 = jidctint.c (JSIMD_FORCENONE=1) */
static int oj_idct_variant = 0;
void oj_set_idct_variant(int v) { oj_idct_variant = v; }
int oj_get_idct_variant(void) { return oj_idct_variant; }

static void oj_idct_block(const int16_t* coef, const uint16_t* q, uint8_t* out, int out_stride)
{
    if (oj_idct_variant)
        oj_idct_block_c(coef, q, out, out_stride);
    else
        oj_idct_block_simd(coef, q, out, out_stride);
}

/* IDCT a whole component into a (bw*8) x (bh*8) plane */
static uint8_t* oj_idct_component(const oj_dec* d, int ci)
{
    const oj_comp* k = &d->comp[ci];
    int pw = k->bw * 8, bx, by;
    uint8_t* plane = (uint8_t*)malloc((size_t)pw * k->bh * 8);
    if (!plane) return NULL;
    for (by = 0; by < k->bh; by++)
        for (bx = 0; bx < k->bw; bx++)
            oj_idct_block(k->coef + ((size_t)by * k->bw + bx) * 64, d->cq[ci], plane + (size_t)by * 8 * pw + bx * 8, pw);
    return plane;
}

/* ---------------------------------------------------------------- upsampling (jdsample.c) */
static inline int oj_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Produce the full-resolution (W x H) plane of component ci from its IDCT plane. */
static uint8_t* oj_upsample(const oj_dec* d, int ci, const uint8_t* plane, int fancy)
{
    const oj_comp* k = &d->comp[ci];
    int W = d->width, H = d->height, pw = k->bw * 8;
    int hx = d->hmax / k->h, vx = d->vmax / k->v; /* expansion factors */
    int x, y;
    uint8_t* out;
    if (d->hmax % k->h || d->vmax % k->v) return NULL; /* fractional sampling: libjpeg errors out too */
    out = (uint8_t*)malloc((size_t)W * H);
    if (!out) return NULL;

    if (hx == 1 && vx == 1) {
        for (y = 0; y < H; y++) memcpy(out + (size_t)y * W, plane + (size_t)y * pw, (size_t)W);
    } else if (hx == 2 && vx == 1 && fancy && k->dw > 2) {
        /* h2v1_fancy_upsample: 3/4 near + 1/4 far, rounding bias alternates 1,2 */
        for (y = 0; y < H; y++) {
            const uint8_t* p = plane + (size_t)y * pw;
            uint8_t* o = out + (size_t)y * W;
            for (x = 0; x < W; x++) {
                int i = x >> 1, v;
                if (x == 0)
                    v = p[0];
                else if (x == 2 * k->dw - 1)
                    v = p[k->dw - 1];
                else if (x & 1)
                    v = (3 * p[i] + p[i + 1] + 2) >> 2;
                else
                    v = (3 * p[i] + p[i - 1] + 1) >> 2;
                o[x] = (uint8_t)v;
            }
        }
    } else if (hx == 2 && vx == 2 && fancy && k->dw > 2) {
        /* h2v2_fancy_upsample: vertical 3:1 first (thiscolsum), then horizontal 3:1, biases 8 / 7 */
        for (y = 0; y < H; y++) {
            int r0 = y >> 1;
            int r1 = oj_clampi((y & 1) ? r0 + 1 : r0 - 1, 0, k->dh - 1);
            const uint8_t* p0 = plane + (size_t)r0 * pw;
            const uint8_t* p1 = plane + (size_t)r1 * pw;
            uint8_t* o = out + (size_t)y * W;
            for (x = 0; x < W; x++) {
                int i = x >> 1, v;
                int cs = 3 * p0[i] + p1[i];
                if (x == 0)
                    v = (cs * 4 + 8) >> 4;
                else if (x == 2 * k->dw - 1)
                    v = (cs * 4 + 7) >> 4;
                else if (x & 1)
                    v = (3 * cs + (3 * p0[i + 1] + p1[i + 1]) + 7) >> 4;
                else
                    v = (3 * cs + (3 * p0[i - 1] + p1[i - 1]) + 8) >> 4;
                o[x] = (uint8_t)v;
            }
        }
    } else if (hx == 1 && vx == 2 && fancy) {
        /* h1v2_fancy_upsample: 3/4 near + 1/4 far vertically; bias 1 for the upper output row, 2 for the lower */
        for (y = 0; y < H; y++) {
            int r0 = y >> 1;
            int r1 = oj_clampi((y & 1) ? r0 + 1 : r0 - 1, 0, k->dh - 1);
            int bias = (y & 1) ? 2 : 1;
            const uint8_t* p0 = plane + (size_t)r0 * pw;
            const uint8_t* p1 = plane + (size_t)r1 * pw;
            uint8_t* o = out + (size_t)y * W;
            for (x = 0; x < W; x++) o[x] = (uint8_t)((3 * p0[x] + p1[x] + bias) >> 2);
        }
    } else {
        /* h2v1 / h2v2 / int_upsample: pixel replication */
        for (y = 0; y < H; y++) {
            const uint8_t* p = plane + (size_t)(y / vx) * pw;
            uint8_t* o = out + (size_t)y * W;
            for (x = 0; x < W; x++) o[x] = p[x / hx];
        }
    }
    return out;
}

/* ---------------------------------------------------------------- colour (jdcolor.c) */
static inline uint8_t oj_clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

static inline void oj_ycc_to_rgb(int y, int cb, int cr, uint8_t* r, uint8_t* g, uint8_t* b)
{
    /* build_ycc_rgb_table: Cr_r = (FIX(1.40200)*x + ONE_HALF) >> 16, Cb_b = (FIX(1.77200)*x + ONE_HALF) >> 16,
       Cr_g = -FIX(0.71414)*x, Cb_g = -FIX(0.34414)*x + ONE_HALF; G uses (Cb_g + Cr_g) >> 16 */
    int cbx = cb - 128, crx = cr - 128;
    *r = oj_clamp8(y + ((91881 * crx + 32768) >> 16));
    *g = oj_clamp8(y + ((-22554 * cbx + 32768 - 46802 * crx) >> 16));
    *b = oj_clamp8(y + ((116130 * cbx + 32768) >> 16));
}

/* ---------------------------------------------------------------- exported decode API */
int oj_read_info(const uint8_t* data, size_t len, oj_info* info)
{
    oj_dec d;
    int rc = oj_parse(&d, data, len, 0), c;
    if (rc != OJ_OK) return rc;
    memset(info, 0, sizeof *info);
    info->width = d.width;
    info->height = d.height;
    info->ncomp = d.ncomp;
    info->sof = d.sof;
    info->colorspace = d.colorspace;
    info->restart_interval = d.restart_interval;
    info->hmax = d.hmax;
    info->vmax = d.vmax;
    for (c = 0; c < d.ncomp; c++) {
        info->h[c] = d.comp[c].h;
        info->v[c] = d.comp[c].v;
        info->bw[c] = d.comp[c].bw;
        info->bh[c] = d.comp[c].bh;
        info->dw[c] = d.comp[c].dw;
        info->dh[c] = d.comp[c].dh;
    }
    return OJ_OK;
}

/* Quantized coefficients of component ci in natural order, [bh][bw][64]; qtab = 64 entries natural order. */
int oj_decode_coefficients(const uint8_t* data, size_t len, int ci, int16_t* coef_out, uint16_t* qtab_out)
{
    oj_dec d;
    int rc = oj_parse(&d, data, len, 1);
    if (rc == OJ_OK) {
        if (ci < 0 || ci >= d.ncomp) {
            rc = OJ_ERR_ARG;
        } else {
            const oj_comp* k = &d.comp[ci];
            memcpy(coef_out, k->coef, (size_t)k->bw * k->bh * 64 * sizeof(int16_t));
            if (qtab_out) memcpy(qtab_out, d.cq[ci], 64 * sizeof(uint16_t));
        }
    }
    oj_free(&d);
    return rc;
}

/* Raw IDCT output of component ci, cropped to its true dw x dh (what P_YUV / P_UNCHANGED hand back). */
int oj_decode_component_plane(const uint8_t* data, size_t len, int ci, uint8_t* out, int stride)
{
    oj_dec d;
    int rc = oj_parse(&d, data, len, 1);
    if (rc == OJ_OK) {
        if (ci < 0 || ci >= d.ncomp) {
            rc = OJ_ERR_ARG;
        } else {
            uint8_t* pl = oj_idct_component(&d, ci);
            int y;
            if (!pl) {
                rc = OJ_ERR_ARG;
            } else {
                for (y = 0; y < d.comp[ci].dh; y++)
                    memcpy(out + (size_t)y * stride, pl + (size_t)y * d.comp[ci].bw * 8, (size_t)d.comp[ci].dw);
                free(pl);
            }
        }
    }
    oj_free(&d);
    return rc;
}

/* Four-component frames, the way the reference's CPU path gets them out of libjpeg-turbo with out_color_space = JCS_CMYK
 * (extensions/libjpeg_turbo/jpeg_mem.cpp:168-172): every component upsampled to full size; a CMYK frame is handed through
 * as it is (jdcolor.c null_convert), a YCCK frame has its first three components turned into R,G,B and complemented, the
 * fourth passed on (jdcolor.c ycck_cmyk_convert: C = 255 - R, M = 255 - G, Y = 255 - B, K = K).  full[c] = dw x dh = width x
 * height planes.  out = 4 bytes per pixel. */
static void oj_cmyk_pixel(const oj_dec* d, uint8_t* const full[4], size_t i, uint8_t cmyk[4])
{
    if (d->colorspace == OJ_CS_YCCK) {
        uint8_t r, g, b;
        oj_ycc_to_rgb(full[0][i], full[1][i], full[2][i], &r, &g, &b);
        cmyk[0] = (uint8_t)(255 - r);
        cmyk[1] = (uint8_t)(255 - g);
        cmyk[2] = (uint8_t)(255 - b);
    } else {
        cmyk[0] = full[0][i];
        cmyk[1] = full[1][i];
        cmyk[2] = full[2][i];
    }
    cmyk[3] = full[3][i];
}

int oj_decode_cmyk(const uint8_t* data, size_t len, int fancy, uint8_t* out, int stride)
{
    oj_dec d;
    uint8_t* full[4] = {NULL, NULL, NULL, NULL};
    int rc = oj_parse(&d, data, len, 1), c, x, y;
    if (rc != OJ_OK) goto done;
    if (d.colorspace != OJ_CS_CMYK && d.colorspace != OJ_CS_YCCK) {
        rc = OJ_ERR_ARG;
        goto done;
    }
    for (c = 0; c < 4; c++) {
        uint8_t* pl = oj_idct_component(&d, c);
        if (!pl) {
            rc = OJ_ERR_ARG;
            goto done;
        }
        full[c] = oj_upsample(&d, c, pl, fancy);
        free(pl);
        if (!full[c]) {
            rc = OJ_ERR_UNSUPPORTED;
            goto done;
        }
    }
    for (y = 0; y < d.height; y++)
        for (x = 0; x < d.width; x++) oj_cmyk_pixel(&d, full, (size_t)y * d.width + x, out + (size_t)y * stride + 4 * x);
done:
    for (c = 0; c < 4; c++) free(full[c]);
    oj_free(&d);
    return rc;
}

/* Full decode to interleaved RGB / BGR (3 bytes per pixel) or a single gray plane. */
int oj_decode(const uint8_t* data, size_t len, int fmt, int fancy, uint8_t* out, int stride)
{
    oj_dec d;
    uint8_t* full[4] = {NULL, NULL, NULL, NULL};
    int rc = oj_parse(&d, data, len, 1), c, x, y, ncolor;
    if (rc != OJ_OK) goto done;
    if (d.colorspace == OJ_CS_CMYK || d.colorspace == OJ_CS_YCCK) {
        /* the reference converts libjpeg's CMYK output itself, per pixel (extensions/libjpeg_turbo/jpeg_mem.cpp:292-337):
         * with an Adobe marker r = k*c/255, without one r = (255-k)*(255-c)/255 (integer division); P_Y = the float expression
         * 0.299f*r + 0.587f*g + 0.114f*b stored into a byte */
        for (c = 0; c < 4; c++) {
            uint8_t* pl = oj_idct_component(&d, c);
            if (!pl) {
                rc = OJ_ERR_ARG;
                goto done;
            }
            full[c] = oj_upsample(&d, c, pl, fancy);
            free(pl);
            if (!full[c]) {
                rc = OJ_ERR_UNSUPPORTED;
                goto done;
            }
        }
        for (y = 0; y < d.height; y++) {
            uint8_t* o = out + (size_t)y * stride;
            for (x = 0; x < d.width; x++) {
                uint8_t q[4];
                int r, g, b;
                oj_cmyk_pixel(&d, full, (size_t)y * d.width + x, q);
                if (d.saw_adobe) {
                    r = (q[3] * q[0]) / 255;
                    g = (q[3] * q[1]) / 255;
                    b = (q[3] * q[2]) / 255;
                } else {
                    r = (255 - q[3]) * (255 - q[0]) / 255;
                    g = (255 - q[3]) * (255 - q[1]) / 255;
                    b = (255 - q[3]) * (255 - q[2]) / 255;
                }
                if (fmt == OJ_FMT_GRAY) {
                    o[x] = (uint8_t)(0.299f * r + 0.587f * g + 0.114f * b);
                } else if (fmt == OJ_FMT_BGR) {
                    o[3 * x] = (uint8_t)b;
                    o[3 * x + 1] = (uint8_t)g;
                    o[3 * x + 2] = (uint8_t)r;
                } else {
                    o[3 * x] = (uint8_t)r;
                    o[3 * x + 1] = (uint8_t)g;
                    o[3 * x + 2] = (uint8_t)b;
                }
            }
        }
        goto done;
    }
    ncolor = (fmt == OJ_FMT_GRAY && d.colorspace != OJ_CS_RGB) ? 1 : d.ncomp;
    for (c = 0; c < ncolor; c++) {
        uint8_t* pl = oj_idct_component(&d, c);
        if (!pl) {
            rc = OJ_ERR_ARG;
            goto done;
        }
        full[c] = oj_upsample(&d, c, pl, fancy);
        free(pl);
        if (!full[c]) {
            rc = OJ_ERR_UNSUPPORTED;
            goto done;
        }
    }
    for (y = 0; y < d.height; y++) {
        uint8_t* o = out + (size_t)y * stride;
        const uint8_t* p0 = full[0] + (size_t)y * d.width;
        const uint8_t* p1 = full[1] ? full[1] + (size_t)y * d.width : NULL;
        const uint8_t* p2 = full[2] ? full[2] + (size_t)y * d.width : NULL;
        for (x = 0; x < d.width; x++) {
            uint8_t r, g, b;
            if (fmt == OJ_FMT_GRAY) {
                if (d.colorspace == OJ_CS_RGB) {
                    /* jdcolor.c rgb_gray_convert: same fixed-point weights as the encoder's Y */
                    o[x] = (uint8_t)((19595 * p0[x] + 38470 * p1[x] + 7471 * p2[x] + 32768) >> 16);
                } else {
                    o[x] = p0[x];
                }
                continue;
            }
            if (d.colorspace == OJ_CS_GRAY) {
                r = g = b = p0[x];
            } else if (d.colorspace == OJ_CS_RGB) {
                r = p0[x];
                g = p1[x];
                b = p2[x];
            } else {
                oj_ycc_to_rgb(p0[x], p1[x], p2[x], &r, &g, &b);
            }
            if (fmt == OJ_FMT_BGR) {
                o[3 * x] = b;
                o[3 * x + 1] = g;
                o[3 * x + 2] = r;
            } else {
                o[3 * x] = r;
                o[3 * x + 1] = g;
                o[3 * x + 2] = b;
            }
        }
    }
done:
    for (c = 0; c < 4; c++) free(full[c]);
    oj_free(&d);
    return rc;
}

/* ================================================================= encoder ================= */

/* T.81 Annex K.1 tables (natural order), jcparam.c std_luminance_quant_tbl / std_chrominance_quant_tbl */
static const uint8_t oj_std_lum_q[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57,
                                         69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64,
                                         81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t oj_std_chr_q[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                         99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                         99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* T.81 Annex K.3 Huffman tables */
static const uint8_t oj_dc_lum_bits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t oj_dc_chr_bits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t oj_dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t oj_ac_lum_bits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t oj_ac_lum_vals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t oj_ac_chr_bits[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t oj_ac_chr_vals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

/* jcparam.c jpeg_quality_scaling + jpeg_add_quant_table (force_baseline) */
void oj_quality_tables(int quality, uint16_t lum[64], uint16_t chr[64])
{
    int scale, i;
    if (quality <= 0) quality = 1;
    if (quality > 100) quality = 100;
    scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (i = 0; i < 64; i++) {
        long a = ((long)oj_std_lum_q[i] * scale + 50L) / 100L;
        long b = ((long)oj_std_chr_q[i] * scale + 50L) / 100L;
        if (a <= 0) a = 1;
        if (a > 255) a = 255;
        if (b <= 0) b = 1;
        if (b > 255) b = 255;
        lum[i] = (uint16_t)a;
        chr[i] = (uint16_t)b;
    }
}

/* jfdctint.c jpeg_fdct_islow: rows then columns, output scaled by 8 */
static void oj_fdct(int32_t* d)
{
    int32_t t0, t1, t2, t3, t4, t5, t6, t7, t10, t11, t12, t13, z1, z2, z3, z4, z5;
    int i;
    int32_t* p = d;
    for (i = 0; i < 8; i++, p += 8) {
        t0 = p[0] + p[7]; t7 = p[0] - p[7];
        t1 = p[1] + p[6]; t6 = p[1] - p[6];
        t2 = p[2] + p[5]; t5 = p[2] - p[5];
        t3 = p[3] + p[4]; t4 = p[3] - p[4];
        t10 = t0 + t3; t13 = t0 - t3; t11 = t1 + t2; t12 = t1 - t2;
        p[0] = (t10 + t11) * 4;
        p[4] = (t10 - t11) * 4;
        z1 = (t12 + t13) * F_0_541;
        p[2] = OJ_DESCALE(z1 + t13 * F_0_765, 11);
        p[6] = OJ_DESCALE(z1 + t12 * (-F_1_847), 11);
        z1 = t4 + t7; z2 = t5 + t6; z3 = t4 + t6; z4 = t5 + t7;
        z5 = (z3 + z4) * F_1_175;
        t4 *= F_0_298; t5 *= F_2_053; t6 *= F_3_072; t7 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        p[7] = OJ_DESCALE(t4 + z1 + z3, 11);
        p[5] = OJ_DESCALE(t5 + z2 + z4, 11);
        p[3] = OJ_DESCALE(t6 + z2 + z3, 11);
        p[1] = OJ_DESCALE(t7 + z1 + z4, 11);
    }
    p = d;
    for (i = 0; i < 8; i++, p++) {
        t0 = p[0] + p[56]; t7 = p[0] - p[56];
        t1 = p[8] + p[48]; t6 = p[8] - p[48];
        t2 = p[16] + p[40]; t5 = p[16] - p[40];
        t3 = p[24] + p[32]; t4 = p[24] - p[32];
        t10 = t0 + t3; t13 = t0 - t3; t11 = t1 + t2; t12 = t1 - t2;
        p[0] = OJ_DESCALE(t10 + t11, 2);
        p[32] = OJ_DESCALE(t10 - t11, 2);
        z1 = (t12 + t13) * F_0_541;
        p[16] = OJ_DESCALE(z1 + t13 * F_0_765, 15);
        p[48] = OJ_DESCALE(z1 + t12 * (-F_1_847), 15);
        z1 = t4 + t7; z2 = t5 + t6; z3 = t4 + t6; z4 = t5 + t7;
        z5 = (z3 + z4) * F_1_175;
        t4 *= F_0_298; t5 *= F_2_053; t6 *= F_3_072; t7 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        p[56] = OJ_DESCALE(t4 + z1 + z3, 15);
        p[40] = OJ_DESCALE(t5 + z2 + z4, 15);
        p[24] = OJ_DESCALE(t6 + z2 + z3, 15);
        p[8] = OJ_DESCALE(t7 + z1 + z4, 15);
    }
}

/*
 * Forward path up to quantized coefficients (jccolor.c rgb_ycc_convert, jcprepct.c edge expansion,
 * jcsample.c h2v1/h2v2/fullsize downsample, jfdctint.c, jcdctmgr.c quantize).
 *   rgb        interleaved R,G,B, `stride` bytes per row
 *   hs/vs      luma sampling factors (chroma is 1x1): (1,1)=4:4:4 (2,1)=4:2:2 (2,2)=4:2:0 (1,2)=4:4:0 (4,1)=4:1:1
 *   ncomp      1 (gray: Y only) or 3
 *   coef[c]    out, [bh][bw][64] natural order, sizes as oj_enc_geometry reports
 */
void oj_enc_geometry(int w, int h, int ncomp, int hs, int vs, int32_t bw[3], int32_t bh[3])
{
    int mcux = (w + 8 * hs - 1) / (8 * hs), mcuy = (h + 8 * vs - 1) / (8 * vs), c;
    for (c = 0; c < ncomp; c++) {
        bw[c] = mcux * (c == 0 ? hs : 1);
        bh[c] = mcuy * (c == 0 ? vs : 1);
    }
    if (ncomp == 1) {
        bw[0] = (w + 7) / 8;
        bh[0] = (h + 7) / 8;
    }
}

int oj_forward(const uint8_t* rgb, int stride, int w, int h, int ncomp, int hs, int vs, const uint16_t* qlum, const uint16_t* qchr,
               int16_t* coef0, int16_t* coef1, int16_t* coef2)
{
    int32_t bw[3], bh[3];
    int16_t* coef[3];
    uint8_t* full[3] = {NULL, NULL, NULL};
    int c, x, y, pw, ph, rc = OJ_OK;
    coef[0] = coef0; coef[1] = coef1; coef[2] = coef2;
    if (ncomp == 1) hs = vs = 1;
    oj_enc_geometry(w, h, ncomp, hs, vs, bw, bh);
    /* full-resolution component planes padded to luma block grid by edge replication (jcprepct.c expand_bottom_edge,
       jcsample.c expand_right_edge) */
    pw = bw[0] * 8;
    ph = bh[0] * 8;
    for (c = 0; c < ncomp; c++) {
        full[c] = (uint8_t*)malloc((size_t)pw * ph);
        if (!full[c]) { rc = OJ_ERR_ARG; goto done; }
    }
    for (y = 0; y < ph; y++) {
        int sy = y < h ? y : h - 1;
        const uint8_t* row = rgb + (size_t)sy * stride;
        for (x = 0; x < pw; x++) {
            int sx = x < w ? x : w - 1;
            int r = row[3 * sx], g = row[3 * sx + 1], b = row[3 * sx + 2];
            full[0][(size_t)y * pw + x] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
            if (ncomp == 3) {
                full[1][(size_t)y * pw + x] = (uint8_t)((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16);
                full[2][(size_t)y * pw + x] = (uint8_t)((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16);
            }
        }
    }
    /* NOTE on vertical padding: libjpeg pads the *input* rows only up to a multiple of vmax (row group), then
       pads the *downsampled* rows of each component up to the block row by replicating the last downsampled row.
       Because downsampling is a per-row-group operation and the replicated input rows are copies of the last
       real row, replicating input rows all the way to ph gives the same downsampled values for the row groups
       that contain real rows; for fully-padded row groups it yields downsample(last row, last row) which can
       differ from replicate(last downsampled row) when the last real row group mixes two different rows.  So
       handle it the libjpeg way: mark rows beyond ceil(h/vs)*vs and fill them after downsampling. */
    for (c = 0; c < ncomp; c++) {
        int ch = c == 0 ? 1 : hs, cv = c == 0 ? 1 : vs; /* downsample factors */
        int cw8 = bw[c] * 8, chh8 = bh[c] * 8;
        int real_rows = c == 0 ? ((h + vs - 1) / vs) * vs : (h + vs - 1) / vs; /* rows produced from real row groups */
        uint8_t* ds = (uint8_t*)malloc((size_t)cw8 * chh8);
        int bx, by;
        if (!ds) { rc = OJ_ERR_ARG; goto done; }
        if (real_rows > chh8) real_rows = chh8;
        for (y = 0; y < real_rows; y++) {
            for (x = 0; x < cw8; x++) {
                int v;
                if (ch == 1 && cv == 1) {
                    v = full[c][(size_t)y * pw + x];
                } else if (ch == 2 && cv == 1) {
                    const uint8_t* p = full[c] + (size_t)y * pw + 2 * x;
                    v = (p[0] + p[1] + (x & 1)) >> 1; /* bias 0,1,0,1 */
                } else if (ch == 2 && cv == 2) {
                    const uint8_t* p = full[c] + (size_t)(2 * y) * pw + 2 * x;
                    v = (p[0] + p[1] + p[pw] + p[pw + 1] + 1 + (x & 1)) >> 2; /* bias 1,2,1,2 */
                } else {
                    /* int_downsample: box average, rounding at numpix/2 */
                    int sum = 0, i, j, n = ch * cv;
                    for (j = 0; j < cv; j++)
                        for (i = 0; i < ch; i++) sum += full[c][(size_t)(cv * y + j) * pw + ch * x + i];
                    v = (sum + n / 2) / n;
                }
                ds[(size_t)y * cw8 + x] = (uint8_t)v;
            }
        }
        for (y = real_rows; y < chh8; y++) memcpy(ds + (size_t)y * cw8, ds + (size_t)(real_rows - 1) * cw8, (size_t)cw8);
        {
            /* real blocks: width_in_blocks x height_in_blocks (jpeglib compptr->width_in_blocks) */
            int dwc = c == 0 ? w : (w + hs - 1) / hs, dhc = c == 0 ? h : (h + vs - 1) / vs;
            int wib = (dwc + 7) / 8, hib = (dhc + 7) / 8;
            int mh = (c == 0 && ncomp == 3) ? hs : 1; /* blocks of this component per MCU, horizontally */
            for (by = 0; by < hib; by++)
                for (bx = 0; bx < wib; bx++) {
                    int32_t blk[64];
                    const uint16_t* q = c == 0 ? qlum : qchr;
                    int16_t* o = coef[c] + ((size_t)by * bw[c] + bx) * 64;
                    int i;
                    for (y = 0; y < 8; y++)
                        for (x = 0; x < 8; x++) blk[y * 8 + x] = (int32_t)ds[(size_t)(by * 8 + y) * cw8 + bx * 8 + x] - 128;
                    oj_fdct(blk);
                    for (i = 0; i < 64; i++) {
                        int32_t qv = (int32_t)q[i] * 8, v = blk[i];
                        if (v < 0) {
                            v = -v;
                            v += qv >> 1;
                            v = v >= qv ? v / qv : 0;
                            v = -v;
                        } else {
                            v += qv >> 1;
                            v = v >= qv ? v / qv : 0;
                        }
                        o[i] = (int16_t)v;
                    }
                }
            /* dummy blocks that complete the last MCU column / row (jccoefct.c compress_data): AC = 0,
               DC = DC of the preceding block in MCU order */
            for (by = 0; by < hib; by++)
                for (bx = wib; bx < bw[c]; bx++) {
                    int16_t* o = coef[c] + ((size_t)by * bw[c] + bx) * 64;
                    memset(o, 0, 64 * sizeof(int16_t));
                    o[0] = o[-64];
                }
            for (by = hib; by < bh[c]; by++)
                for (bx = 0; bx < bw[c]; bx++) {
                    int16_t* o = coef[c] + ((size_t)by * bw[c] + bx) * 64;
                    int last_in_mcu = (bx / mh) * mh + mh - 1;
                    memset(o, 0, 64 * sizeof(int16_t));
                    o[0] = coef[c][((size_t)(by - 1) * bw[c] + last_in_mcu) * 64];
                }
        }
        free(ds);
    }
done:
    for (c = 0; c < 3; c++) free(full[c]);
    return rc;
}

/* Pre-subsampled planar YCbCr input (NVIMGCODEC_SAMPLEFORMAT_P_YUV, which the reference's GPU encoder hands to nvjpegEncodeYUV,
 * extensions/nvjpeg/cuda_encoder.cpp:362-368): the planes ARE the components -- level shift, FDCT, quantize, nothing else.
 * Plane c holds ceil(w / hs_c) x ceil(h / vs_c) samples; blocks that reach past a plane's edge replicate its last column /
 * row (the rule libjpeg applies to what it downsamples; nvJPEG's own rule is not documented in the reference, so this
 * function is "parity unpinned" beyond sizes where no padding occurs).  Output as oj_forward: natural-order blocks over the
 * MCU-padded grid, dummy blocks filled the jccoefct.c way. */
int oj_forward_planes(const uint8_t* p0, int s0, const uint8_t* p1, int s1, const uint8_t* p2, int s2, int w, int h, int hs, int vs,
                      const uint16_t* qlum, const uint16_t* qchr, int16_t* coef0, int16_t* coef1, int16_t* coef2)
{
    int32_t bw[3], bh[3];
    int16_t* coef[3];
    const uint8_t* plane[3];
    int stride[3], c, x, y, bx, by;
    coef[0] = coef0; coef[1] = coef1; coef[2] = coef2;
    plane[0] = p0; plane[1] = p1; plane[2] = p2;
    stride[0] = s0; stride[1] = s1; stride[2] = s2;
    oj_enc_geometry(w, h, 3, hs, vs, bw, bh);
    for (c = 0; c < 3; c++) {
        int dwc = c == 0 ? w : (w + hs - 1) / hs, dhc = c == 0 ? h : (h + vs - 1) / vs;
        int wib = (dwc + 7) / 8, hib = (dhc + 7) / 8;
        int mh = c == 0 ? hs : 1;
        const uint16_t* q = c == 0 ? qlum : qchr;
        for (by = 0; by < hib; by++)
            for (bx = 0; bx < wib; bx++) {
                int32_t blk[64];
                int16_t* o = coef[c] + ((size_t)by * bw[c] + bx) * 64;
                int i;
                for (y = 0; y < 8; y++)
                    for (x = 0; x < 8; x++) {
                        int sy = by * 8 + y, sx = bx * 8 + x;
                        if (sy > dhc - 1) sy = dhc - 1;
                        if (sx > dwc - 1) sx = dwc - 1;
                        blk[y * 8 + x] = (int32_t)plane[c][(size_t)sy * stride[c] + sx] - 128;
                    }
                oj_fdct(blk);
                for (i = 0; i < 64; i++) {
                    int32_t qv = (int32_t)q[i] * 8, v = blk[i];
                    if (v < 0) {
                        v = -v;
                        v += qv >> 1;
                        v = v >= qv ? v / qv : 0;
                        v = -v;
                    } else {
                        v += qv >> 1;
                        v = v >= qv ? v / qv : 0;
                    }
                    o[i] = (int16_t)v;
                }
            }
        for (by = 0; by < hib; by++)
            for (bx = wib; bx < bw[c]; bx++) {
                int16_t* o = coef[c] + ((size_t)by * bw[c] + bx) * 64;
                memset(o, 0, 64 * sizeof(int16_t));
                o[0] = o[-64];
            }
        for (by = hib; by < bh[c]; by++)
            for (bx = 0; bx < bw[c]; bx++) {
                int16_t* o = coef[c] + ((size_t)by * bw[c] + bx) * 64;
                int last_in_mcu = (bx / mh) * mh + mh - 1;
                memset(o, 0, 64 * sizeof(int16_t));
                o[0] = coef[c][((size_t)(by - 1) * bw[c] + last_in_mcu) * 64];
            }
    }
    return OJ_OK;
}

/* ---------------------------------------------------------------- entropy encoder + JFIF writer */
typedef struct {
    uint8_t* p;
    size_t cap, n;
    uint32_t acc;
    int nbits;
    int overflow;
} oj_out;

static void oj_put(oj_out* o, int b)
{
    if (o->n < o->cap)
        o->p[o->n] = (uint8_t)b;
    else
        o->overflow = 1;
    o->n++;
}
static void oj_put16(oj_out* o, int v) { oj_put(o, v >> 8); oj_put(o, v & 255); }

static void oj_emit(oj_out* o, unsigned code, int size)
{
    o->acc = (o->acc << size) | (code & ((1u << size) - 1));
    o->nbits += size;
    while (o->nbits >= 8) {
        int b = (o->acc >> (o->nbits - 8)) & 255;
        oj_put(o, b);
        if (b == 0xFF) oj_put(o, 0);
        o->nbits -= 8;
    }
}
static void oj_flush_bits(oj_out* o)
{
    if (o->nbits > 0) oj_emit(o, 0x7F, 8 - o->nbits); /* pad with 1s (jchuff.c flush_bits) */
    o->acc = 0;
    o->nbits = 0;
}

typedef struct { uint16_t code[256]; uint8_t size[256]; } oj_ehuff;

static void oj_make_ehuff(const uint8_t* bits, const uint8_t* vals, oj_ehuff* t)
{
    int l, i, k = 0;
    unsigned code = 0;
    memset(t, 0, sizeof *t);
    for (l = 1; l <= 16; l++) {
        for (i = 0; i < bits[l]; i++, k++) {
            t->code[vals[k]] = (uint16_t)code++;
            t->size[vals[k]] = (uint8_t)l;
        }
        code <<= 1;
    }
}

static int oj_nbits(int v)
{
    int n = 0;
    if (v < 0) v = -v;
    while (v) { n++; v >>= 1; }
    return n;
}

static void oj_encode_block(oj_out* o, const int16_t* blk, int* pred, const oj_ehuff* dc, const oj_ehuff* ac)
{
    int diff = blk[0] - *pred, t = diff, n, k, r = 0;
    *pred = blk[0];
    if (t < 0) { t = -t; diff--; }
    n = oj_nbits(t);
    oj_emit(o, dc->code[n], dc->size[n]);
    if (n) oj_emit(o, (unsigned)diff, n);
    for (k = 1; k < 64; k++) {
        int v = blk[oj_zigzag[k]], v2;
        if (v == 0) { r++; continue; }
        while (r > 15) { oj_emit(o, ac->code[0xF0], ac->size[0xF0]); r -= 16; }
        v2 = v;
        if (v < 0) { v = -v; v2--; }
        n = oj_nbits(v);
        oj_emit(o, ac->code[(r << 4) + n], ac->size[(r << 4) + n]);
        oj_emit(o, (unsigned)v2, n);
        r = 0;
    }
    if (r > 0) oj_emit(o, ac->code[0], ac->size[0]);
}

static void oj_write_dht(oj_out* o, int tc_th, const uint8_t* bits, const uint8_t* vals)
{
    int n = 0, i;
    for (i = 1; i <= 16; i++) n += bits[i];
    oj_put16(o, 0xFFC4);
    oj_put16(o, 2 + 1 + 16 + n);
    oj_put(o, tc_th);
    for (i = 1; i <= 16; i++) oj_put(o, bits[i]);
    for (i = 0; i < n; i++) oj_put(o, vals[i]);
}

/*
 * Baseline sequential encode with the Annex-K Huffman tables, libjpeg's marker order (jcmarker.c):
 * SOI, APP0(JFIF 1.01), DQT(0), DQT(1), SOF0, DHT x4 (x2 for gray), [DRI], SOS, data, EOI.
 * Returns number of bytes produced (> cap means the buffer was too small), or <0 on error.
 * restart_interval is in MCUs (0 = none).
 */
long oj_encode(const uint8_t* rgb, int stride, int w, int h, int ncomp, int hs, int vs, int quality, int restart_interval,
               uint8_t* out, size_t cap)
{
    uint16_t ql[64], qc[64];
    int32_t bw[3], bh[3];
    int16_t* coef[3] = {NULL, NULL, NULL};
    oj_out o;
    oj_ehuff dcl, dcc, acl, acc;
    int c, i, mx, my, mcux, mcuy, pred[3] = {0, 0, 0}, rst = 0, left;
    long ret;
    if (ncomp != 1 && ncomp != 3) return OJ_ERR_ARG;
    if (ncomp == 1) hs = vs = 1;
    oj_quality_tables(quality, ql, qc);
    oj_enc_geometry(w, h, ncomp, hs, vs, bw, bh);
    for (c = 0; c < ncomp; c++) {
        coef[c] = (int16_t*)malloc((size_t)bw[c] * bh[c] * 64 * sizeof(int16_t));
        if (!coef[c]) { ret = OJ_ERR_ARG; goto done; }
    }
    if (oj_forward(rgb, stride, w, h, ncomp, hs, vs, ql, qc, coef[0], coef[1], coef[2]) != OJ_OK) { ret = OJ_ERR_ARG; goto done; }
    memset(&o, 0, sizeof o);
    o.p = out;
    o.cap = cap;
    oj_put16(&o, 0xFFD8);
    oj_put16(&o, 0xFFE0); oj_put16(&o, 16);
    oj_put(&o, 'J'); oj_put(&o, 'F'); oj_put(&o, 'I'); oj_put(&o, 'F'); oj_put(&o, 0);
    oj_put16(&o, 0x0101); oj_put(&o, 0); oj_put16(&o, 1); oj_put16(&o, 1); oj_put(&o, 0); oj_put(&o, 0);
    for (c = 0; c < (ncomp == 3 ? 2 : 1); c++) {
        oj_put16(&o, 0xFFDB); oj_put16(&o, 67); oj_put(&o, c);
        for (i = 0; i < 64; i++) oj_put(&o, (c ? qc : ql)[oj_zigzag[i]]);
    }
    oj_put16(&o, 0xFFC0); oj_put16(&o, 8 + 3 * ncomp); oj_put(&o, 8); oj_put16(&o, h); oj_put16(&o, w); oj_put(&o, ncomp);
    for (c = 0; c < ncomp; c++) {
        oj_put(&o, c + 1);
        oj_put(&o, c == 0 ? ((hs << 4) | vs) : 0x11);
        oj_put(&o, c == 0 ? 0 : 1);
    }
    oj_write_dht(&o, 0x00, oj_dc_lum_bits, oj_dc_vals);
    oj_write_dht(&o, 0x10, oj_ac_lum_bits, oj_ac_lum_vals);
    if (ncomp == 3) {
        oj_write_dht(&o, 0x01, oj_dc_chr_bits, oj_dc_vals);
        oj_write_dht(&o, 0x11, oj_ac_chr_bits, oj_ac_chr_vals);
    }
    if (restart_interval) { oj_put16(&o, 0xFFDD); oj_put16(&o, 4); oj_put16(&o, restart_interval); }
    oj_put16(&o, 0xFFDA); oj_put16(&o, 6 + 2 * ncomp); oj_put(&o, ncomp);
    for (c = 0; c < ncomp; c++) { oj_put(&o, c + 1); oj_put(&o, c == 0 ? 0x00 : 0x11); }
    oj_put(&o, 0); oj_put(&o, 63); oj_put(&o, 0);
    oj_make_ehuff(oj_dc_lum_bits, oj_dc_vals, &dcl);
    oj_make_ehuff(oj_dc_chr_bits, oj_dc_vals, &dcc);
    oj_make_ehuff(oj_ac_lum_bits, oj_ac_lum_vals, &acl);
    oj_make_ehuff(oj_ac_chr_bits, oj_ac_chr_vals, &acc);
    mcux = ncomp == 1 ? bw[0] : bw[0] / hs;
    mcuy = ncomp == 1 ? bh[0] : bh[0] / vs;
    left = restart_interval;
    for (my = 0; my < mcuy; my++)
        for (mx = 0; mx < mcux; mx++) {
            if (restart_interval && left == 0) {
                oj_flush_bits(&o);
                oj_put16(&o, 0xFFD0 + rst);
                rst = (rst + 1) & 7;
                pred[0] = pred[1] = pred[2] = 0;
                left = restart_interval;
            }
            for (c = 0; c < ncomp; c++) {
                int ch = (c == 0 && ncomp == 3) ? hs : 1, cv = (c == 0 && ncomp == 3) ? vs : 1, bx, by;
                for (by = 0; by < cv; by++)
                    for (bx = 0; bx < ch; bx++)
                        oj_encode_block(&o, coef[c] + ((size_t)(my * cv + by) * bw[c] + mx * ch + bx) * 64, &pred[c],
                                        c == 0 ? &dcl : &dcc, c == 0 ? &acl : &acc);
            }
            left--;
        }
    oj_flush_bits(&o);
    oj_put16(&o, 0xFFD9);
    ret = (long)o.n;
done:
    for (c = 0; c < 3; c++) free(coef[c]);
    return ret;
}
