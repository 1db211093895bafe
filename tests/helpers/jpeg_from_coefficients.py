"""Test helper: a baseline JPEG file written from CHOSEN quantized coefficients and quantization tables (Annex K Huffman tables),
so that tests can put exact values where the kernels' arithmetic changes flavour (24-bit / 32-bit multipliers, packed int16).
Plain Python + numpy; small pictures only."""
import numpy as np

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
          57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]

# ITU-T T.81 Annex K.3 typical Huffman tables
DC_LUMA = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_CHROMA = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_LUMA = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
           [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
            0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
            0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
            0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
            0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
            0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
            0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
            0xfa])
AC_CHROMA = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
             [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
              0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
              0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
              0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
              0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
              0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
              0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
              0xfa])


def _codes(bits, vals):
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            table[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return table


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _magnitude(v):
    a = abs(int(v))
    nb = a.bit_length()
    return nb, (int(v) if v >= 0 else int(v) - 1) & ((1 << nb) - 1)


def write_baseline(width, height, sampling, coefficients, qtables):
    """sampling: [(h, v)] per component; coefficients[c]: int array [blocks_h][blocks_w][64] in natural (row-major) order over the
    MCU-padded block grid, DC values absolute; qtables[c]: 64 quantizers in natural order (values 1..255; any value above 255 in a
    table makes it a 16-bit table and the frame an extended-sequential one, SOF1).  Component 0 uses the luminance Huffman tables,
    the others the chrominance ones.  DC values may leave int16 as long as successive differences fit 11 bits: libjpeg keeps the
    predictor in an int and stores its low 16 bits in the block.  Returns the file as bytes."""
    ncomp = len(sampling)
    hmax, vmax = max(h for h, _ in sampling), max(v for _, v in sampling)
    mcus_x, mcus_y = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = bytearray(b"\xff\xd8")
    wide = False
    for c in range(ncomp):
        q = [int(qtables[c][ZIGZAG[k]]) for k in range(64)]
        assert all(1 <= x <= 65535 for x in q)
        if max(q) > 255:
            wide = True
            out += b"\xff\xdb" + (131).to_bytes(2, "big") + bytes([0x10 | c]) + b"".join(x.to_bytes(2, "big") for x in q)
        else:
            out += b"\xff\xdb" + (67).to_bytes(2, "big") + bytes([c]) + bytes(q)
    out += (b"\xff\xc1" if wide else b"\xff\xc0") + (8 + 3 * ncomp).to_bytes(2, "big") + b"\x08" + height.to_bytes(2, "big") + width.to_bytes(2, "big") + bytes([ncomp])
    for c, (h, v) in enumerate(sampling):
        out += bytes([c + 1, (h << 4) | v, c])
    tables = [(0x00, DC_LUMA), (0x10, AC_LUMA)] + ([(0x01, DC_CHROMA), (0x11, AC_CHROMA)] if ncomp > 1 else [])
    for ident, (bits, vals) in tables:
        out += b"\xff\xc4" + (19 + len(vals)).to_bytes(2, "big") + bytes([ident]) + bytes(bits) + bytes(vals)
    out += b"\xff\xda" + (6 + 2 * ncomp).to_bytes(2, "big") + bytes([ncomp])
    for c in range(ncomp):
        out += bytes([c + 1, 0x00 if c == 0 else 0x11])
    out += b"\x00\x3f\x00"
    dc_codes = [_codes(*DC_LUMA)] + [_codes(*DC_CHROMA)] * (ncomp - 1)
    ac_codes = [_codes(*AC_LUMA)] + [_codes(*AC_CHROMA)] * (ncomp - 1)
    bw = _Bits()
    pred = [0] * ncomp
    for my in range(mcus_y):
        for mx in range(mcus_x):
            for c, (h, v) in enumerate(sampling):
                for by in range(v):
                    for bx in range(h):
                        blk = coefficients[c][my * v + by][mx * h + bx]
                        diff = int(blk[0]) - pred[c]
                        pred[c] = int(blk[0])
                        nb, bits = _magnitude(diff)
                        assert nb <= 11, "DC difference beyond category 11"
                        bw.put(*dc_codes[c][nb])
                        if nb:
                            bw.put(bits, nb)
                        run = 0
                        for k in range(1, 64):
                            val = int(blk[ZIGZAG[k]])
                            if val == 0:
                                run += 1
                                continue
                            while run > 15:
                                bw.put(*ac_codes[c][0xF0])
                                run -= 16
                            nb, bits = _magnitude(val)
                            assert nb <= 10
                            bw.put(*ac_codes[c][(run << 4) | nb])
                            bw.put(bits, nb)
                            run = 0
                        if run:
                            bw.put(*ac_codes[c][0x00])
    bw.flush()
    out += bw.out + b"\xff\xd9"
    return bytes(out)


def block_grid(width, height, sampling):
    """[(blocks_h, blocks_w)] per component of the MCU-padded grid write_baseline() expects."""
    hmax, vmax = max(h for h, _ in sampling), max(v for _, v in sampling)
    mcus_x, mcus_y = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    return [(mcus_y * v, mcus_x * h) for h, v in sampling]


def random_coefficients(rng, width, height, sampling, extreme, dense=6, small=3, dc=60):
    """Blocks with a handful of small coefficients and up to three of magnitude `extreme` at random AC positions and signs."""
    res = []
    for bh, bwid in block_grid(width, height, sampling):
        a = np.zeros((bh, bwid, 64), dtype=np.int32)
        a[..., 0] = rng.integers(-dc, dc, size=(bh, bwid))
        for _ in range(dense):
            pos = rng.integers(1, 64, size=(bh, bwid))
            val = rng.integers(-small, small + 1, size=(bh, bwid))
            np.put_along_axis(a, pos[..., None], val[..., None], axis=2)
        for _ in range(3):
            pos = rng.integers(1, 64, size=(bh, bwid))
            val = rng.choice([-extreme, extreme, 0], size=(bh, bwid))
            np.put_along_axis(a, pos[..., None], val[..., None], axis=2)
        res.append(a)
    return res


def write_progressive(width, height, sampling, coefficients, qtables, script):
    """A progressive (SOF2) file with a CHOSEN scan script, for the scan layouts libjpeg's default script does not produce (DC scans of one
    component, DC scans of some components, AC bands cut anywhere, refinement passes over parts of a band).  script: list of
    ("dc", [component, ...], ah, al) and ("ac", component, ss, se, ah, al) -- ah / al may be left out (0, 0: spectral selection only).  The
    caller keeps the progression consistent (T.81 G.1.1.1.1: a first pass with ah = 0 before refinements, each with ah = the previous al and
    al = ah - 1; a component's DC before its AC).  Coding follows jcphuff.c with end-of-band runs of ONE block (every block closes its band
    itself).  Other arguments as write_baseline (8-bit tables only).  Interleaved scans walk the MCU-padded grid MCU by MCU; a scan of one
    component walks its REAL blocks (ceil(samples / 8) each way) in raster order (T.81 A.2.2).  Returns the file as bytes."""
    ncomp = len(sampling)
    hmax, vmax = max(h for h, _ in sampling), max(v for _, v in sampling)
    mcus_x, mcus_y = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = bytearray(b"\xff\xd8")
    for c in range(ncomp):
        q = [int(qtables[c][ZIGZAG[k]]) for k in range(64)]
        assert all(1 <= x <= 255 for x in q)
        out += b"\xff\xdb" + (67).to_bytes(2, "big") + bytes([c]) + bytes(q)
    out += b"\xff\xc2" + (8 + 3 * ncomp).to_bytes(2, "big") + b"\x08" + height.to_bytes(2, "big") + width.to_bytes(2, "big") + bytes([ncomp])
    for c, (h, v) in enumerate(sampling):
        out += bytes([c + 1, (h << 4) | v, c])
    tables = [(0x00, DC_LUMA), (0x10, AC_LUMA)] + ([(0x01, DC_CHROMA), (0x11, AC_CHROMA)] if ncomp > 1 else [])
    for ident, (bits, vals) in tables:
        out += b"\xff\xc4" + (19 + len(vals)).to_bytes(2, "big") + bytes([ident]) + bytes(bits) + bytes(vals)
    dc_codes = [_codes(*DC_LUMA)] + [_codes(*DC_CHROMA)] * (ncomp - 1)
    ac_codes = [_codes(*AC_LUMA)] + [_codes(*AC_CHROMA)] * (ncomp - 1)

    def real_blocks(c):
        h, v = sampling[c]
        cw, ch = -(-width * h // hmax), -(-height * v // vmax)  # the component's samples
        return -(-cw // 8), -(-ch // 8)

    for entry in script:
        bw = _Bits()
        if entry[0] == "dc":
            comps = list(entry[1])
            ah, al = (entry[2], entry[3]) if len(entry) > 2 else (0, 0)
            out += b"\xff\xda" + (6 + 2 * len(comps)).to_bytes(2, "big") + bytes([len(comps)])
            for c in comps:
                out += bytes([c + 1, 0x00 if c == 0 else 0x10])
            out += bytes([0, 0, (ah << 4) | al])
            pred = {c: 0 for c in comps}

            def dc(c, blk):
                if ah:  # refinement: the next bit of the value (jcphuff.c encode_mcu_DC_refine)
                    bw.put((int(blk[0]) >> al) & 1, 1)
                    return
                v = int(blk[0]) >> al  # arithmetic shift (encode_mcu_DC_first)
                diff = v - pred[c]
                pred[c] = v
                nb, bits = _magnitude(diff)
                assert nb <= 11
                bw.put(*dc_codes[c][nb])
                if nb:
                    bw.put(bits, nb)

            if len(comps) == 1:
                c = comps[0]
                nbx, nby = real_blocks(c)
                for y in range(nby):
                    for x in range(nbx):
                        dc(c, coefficients[c][y][x])
            else:
                for my in range(mcus_y):
                    for mx in range(mcus_x):
                        for c in comps:
                            h, v = sampling[c]
                            for by in range(v):
                                for bx in range(h):
                                    dc(c, coefficients[c][my * v + by][mx * h + bx])
        else:
            c, ss, se = entry[1], entry[2], entry[3]
            ah, al = (entry[4], entry[5]) if len(entry) > 4 else (0, 0)
            assert 1 <= ss <= se <= 63
            out += b"\xff\xda" + (8).to_bytes(2, "big") + bytes([1, c + 1, 0x00 if c == 0 else 0x11, ss, se, (ah << 4) | al])
            nbx, nby = real_blocks(c)
            for y in range(nby):
                for x in range(nbx):
                    blk = coefficients[c][y][x]
                    if ah == 0:  # first pass of the band (encode_mcu_AC_first): the values shifted down by al, magnitude first
                        run = 0
                        for k in range(ss, se + 1):
                            val = int(blk[ZIGZAG[k]])
                            mag = abs(val) >> al
                            if mag == 0:
                                run += 1
                                continue
                            while run > 15:
                                bw.put(*ac_codes[c][0xF0])
                                run -= 16
                            nb, bits = _magnitude(mag if val > 0 else -mag)
                            assert nb <= 10
                            bw.put(*ac_codes[c][(run << 4) | nb])
                            bw.put(bits, nb)
                            run = 0
                        if run:
                            bw.put(*ac_codes[c][0x00])  # end of band, a run of one block
                    else:  # refinement (encode_mcu_AC_refine)
                        mags = {k: abs(int(blk[ZIGZAG[k]])) >> al for k in range(ss, se + 1)}
                        eob = max([k for k in mags if mags[k] == 1], default=0)  # the last coefficient that becomes non-zero in this pass
                        run, pending = 0, []
                        for k in range(ss, se + 1):
                            mag = mags[k]
                            if mag == 0:
                                run += 1
                                continue
                            while run > 15 and k <= eob:  # (not if the zeros can be folded into the end of band)
                                bw.put(*ac_codes[c][0xF0])
                                run -= 16
                                for b in pending:
                                    bw.put(b, 1)
                                pending = []
                            if mag > 1:  # non-zero before: its next bit, after the next symbol
                                pending.append(mag & 1)
                                continue
                            bw.put(*ac_codes[c][(run << 4) | 1])
                            bw.put(0 if int(blk[ZIGZAG[k]]) < 0 else 1, 1)
                            for b in pending:
                                bw.put(b, 1)
                            pending, run = [], 0
                        if run > 0 or pending:
                            bw.put(*ac_codes[c][0x00])
                            for b in pending:
                                bw.put(b, 1)
        bw.flush()
        out += bw.out
    out += b"\xff\xd9"
    return bytes(out)
