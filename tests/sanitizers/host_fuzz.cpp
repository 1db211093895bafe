// host_fuzz.cpp -- AddressSanitizer / UBSan harness for the host-side code that reads untrusted bytes: the marker parser, the host
// entropy decoder, and the host emulations of the GPU entropy stage (the kernels' own decode routines).  GPU sanitizers are not
// available on the pool, so this is where memory errors of that code would show; tests/test_sanitizers.py builds and runs it.
// usage: host_fuzz <iterations> <seed> file.jpg...   -- every file as it is, then mutated copies (bit flips, truncation, spliced
// markers); prints a summary line, exits non-zero only if a sanitizer aborts or two decoders disagree on a stream both accept.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <vector>

#include "entropy_decode.h"
#include "gpu_huffman_host.h"
#include "jpeg_syntax.h"
#include "progressive_gpu_host.h"

using namespace hipjpeg;

static long g_parsed = 0, g_decoded = 0, g_emulated = 0, g_mismatch = 0;

static void run_one(const std::vector<uint8_t>& bytes)
{
    // exact-size heap copy: a read one byte past the end lands in ASan's red zone
    std::vector<uint8_t> copy(bytes);
    const uint8_t* data = copy.data();
    const size_t length = copy.size();
    FrameInfo f;
    if (parse_jpeg(data, length, &f) != kParseOk) return;
    g_parsed++;
    if (f.total_blocks() * 128 > (64u << 20)) return;  // forged sizes: the product has its own cap (max_image_samples)
    std::vector<int16_t> a(f.total_blocks() * 64), b(f.total_blocks() * 64);
    int16_t *pa[4] = {nullptr, nullptr, nullptr, nullptr}, *pb[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t off = 0;
    for (int c = 0; c < f.ncomp; c++) {
        pa[c] = a.data() + off;
        pb[c] = b.data() + off;
        off += (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h * 64;
    }
    const int host = decode_coefficients(data, length, f, pa);
    g_decoded += host == kEntropyOk;
    if (sparse_staging_applies(f)) {
        // the zero-run-compressed writer (what crosses PCIe for host-decoded pictures): same verdict as the dense decoder, and a stream
        // that expands to the dense blocks -- under the sanitizers, on mutated input
        std::vector<uint8_t> stream(sparse_stream_capacity(f));
        size_t bytes = 0;
        const int sp = decode_coefficients_sparse(data, length, f, stream.data(), &bytes);
        if ((sp == kEntropyOk) != (host == kEntropyOk)) {
            g_mismatch++;
            fprintf(stderr, "sparse and dense host decoders disagree on the verdict (%d vs %d)\n", sp, host);
        } else if (sp == kEntropyOk) {
            if (bytes > stream.size()) abort();
            const uint32_t* table = reinterpret_cast<const uint32_t*>(stream.data());
            size_t first = 0;
            for (int c = 0; c < f.ncomp; c++) {
                const size_t nb = (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h;
                for (size_t b2 = 0; b2 < nb; b2++) {
                    int16_t blk[64] = {0};
                    const uint32_t off = table[first + b2];
                    if (off) {
                        if (off + 3 > bytes) abort();
                        const uint8_t* rec = stream.data() + off;
                        const int k = rec[0];
                        if (off + 3 + 3 * (size_t)k > bytes) abort();
                        blk[0] = (int16_t)(rec[1] | (rec[2] << 8));
                        for (int e = 0; e < k; e++) blk[rec[3 + 3 * e] & 63] = (int16_t)(rec[4 + 3 * e] | (rec[5 + 3 * e] << 8));
                    }
                    const size_t by = b2 / f.comp[c].blocks_w, bx = b2 % f.comp[c].blocks_w;
                    const bool coded = off != 0 || (by < (size_t)(f.comp[c].samp_h + 7) / 8 && bx < (size_t)(f.comp[c].samp_w + 7) / 8);
                    if (coded && memcmp(blk, pa[c] + b2 * 64, 128) != 0) {
                        g_mismatch++;
                        fprintf(stderr, "sparse stream differs from the dense blocks (component %d block %zu)\n", c, b2);
                        break;
                    }
                }
                first += nb;
            }
        }
    }
    const bool prog = gpu_progressive_eligible(f);
    if (!prog && !gpu_entropy_eligible(f)) return;
    int passes = 0;
    const int rc = prog ? emulate_gpu_progressive(data, length, f, pb) : emulate_gpu_entropy(data, length, f, pb, &passes);
    g_emulated++;
    if (host == kEntropyOk && rc == 0 && memcmp(a.data(), b.data(), a.size() * 2) != 0) {
        // MCU padding rows of components whose sampling factor is not maximal are never written by either side: compare what is
        const bool real_difference = [&] {
            for (int c = 0; c < f.ncomp; c++)
                for (int by = 0; by < f.comp[c].blocks_h; by++)
                    for (int bx = 0; bx < f.comp[c].blocks_w; bx++) {
                        const size_t o = ((size_t)by * f.comp[c].blocks_w + bx) * 64;
                        if (memcmp(pa[c] + o, pb[c] + o, 128) != 0 && by < (f.comp[c].samp_h + 7) / 8 && bx < (f.comp[c].samp_w + 7) / 8) return true;
                    }
            return false;
        }();
        if (real_difference) g_mismatch++;
    }
}

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    const long iterations = atol(argv[1]);
    std::mt19937 rng((unsigned)atol(argv[2]));
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 3; i < argc; i++) {
        std::ifstream in(argv[i], std::ios::binary);
        seeds.emplace_back(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
        run_one(seeds.back());
    }
    for (long it = 0; it < iterations; it++) {
        std::vector<uint8_t> m = seeds[rng() % seeds.size()];
        if (m.size() < 8) continue;
        switch (rng() % 6) {
        case 0:  // bit flips anywhere
            for (unsigned k = 0, n = 1 + rng() % 4; k < n; k++) m[rng() % m.size()] ^= (uint8_t)(1u << (rng() % 8));
            break;
        case 1:  // truncation
            m.resize(2 + rng() % (m.size() - 2));
            break;
        case 2:  // byte overwrite inside the headers
            m[2 + rng() % std::min<size_t>(m.size() - 2, 700)] = (uint8_t)rng();
            break;
        case 3: {  // a segment length field set to something else
            for (size_t p = 2; p + 4 < m.size() && p < 900; p++)
                if (m[p] == 0xFF && m[p + 1] >= 0xC0 && m[p + 1] != 0xFF && (rng() % 3) == 0) {
                    m[p + 2] = (uint8_t)(rng() % 4 == 0 ? 0 : rng());
                    m[p + 3] = (uint8_t)rng();
                    break;
                }
            break;
        }
        case 4:  // a marker spliced into the entropy-coded data
            if (m.size() > 700) {
                const size_t p = 650 + rng() % (m.size() - 652);
                m[p] = 0xFF;
                m[p + 1] = (uint8_t)(0xC0 + rng() % 0x3F);
            }
            break;
        default:  // the file cut right behind a marker / inside a segment
            for (size_t p = 2; p + 2 < m.size(); p++)
                if (m[p] == 0xFF && m[p + 1] == 0xDA && (rng() % 2)) {
                    m.resize(p + 2 + rng() % 12);
                    break;
                }
        }
        run_one(m);
    }
    printf("host_fuzz: %ld parsed, %ld decoded by the host decoder, %ld through the GPU-algorithm emulation, %ld coefficient mismatches\n", g_parsed,
           g_decoded, g_emulated, g_mismatch);
    return g_mismatch ? 1 : 0;
}
