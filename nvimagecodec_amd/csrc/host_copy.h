// host_copy.h -- copying bitstreams into the pinned staging area.  The destination is written once and next read by the GPU's copy
// over PCIe, never by this CPU: non-temporal stores skip the read-for-ownership of every destination line (a third of the memory
// traffic of a plain memcpy; glibc only switches to them for single copies far larger than one image's bitstream).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace hipjpeg {

#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void stream_copy_avx2(uint8_t* dst, const uint8_t* src, size_t n)
{
    size_t head = (32 - (reinterpret_cast<uintptr_t>(dst) & 31)) & 31;
    if (head > n) head = n;
    memcpy(dst, src, head);
    dst += head;
    src += head;
    n -= head;
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 96), d);
    }
    _mm_sfence();  // the stores are globally visible before the caller queues the copy that reads them
    memcpy(dst + i, src + i, n - i);
}
#endif

inline void copy_to_staging(uint8_t* dst, const uint8_t* src, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 4096) {
        stream_copy_avx2(dst, src, n);
        return;
    }
#endif
    memcpy(dst, src, n);
}

}  // namespace hipjpeg
