// diagnostics.h -- profiler ranges and fault injection points of the host code.
//
// Ranges: the reference brackets its stages with NVTX scoped ranges (extensions/nvjpeg/cuda_decoder.cpp:415,502,528,547,
// src/decoder_worker.cpp:254); here the same sites push roctx ranges, which rocprofv3 --marker-trace shows next to the
// kernels.  The roctx library is looked up at run time (librocprofiler-sdk-roctx, else libroctx64): without it, or with
// HIPJPEG_NO_ROCTX set, a range costs one predictable branch.
//
// Fault points: tests arm a named site (hipjpegTestSetFault) so that its n-th passage throws std::runtime_error; the
// boundary code must then still report every sample exactly once and stay usable (tests/test_gpu_plugin.py).
#pragma once

namespace hipjpeg {

void range_push(const char* name);
void range_pop();

struct ScopedRange {
    explicit ScopedRange(const char* name) { range_push(name); }
    ~ScopedRange() { range_pop(); }
    ScopedRange(const ScopedRange&) = delete;
    ScopedRange& operator=(const ScopedRange&) = delete;
};

// Throws std::runtime_error when `site` is armed and its countdown reaches zero; a no-op (one relaxed load) otherwise.
void fault_point(const char* site);
// Arms `site` (nullptr / "" disarms): the countdown-th passage from now throws, once.
void set_fault(const char* site, int countdown);

}  // namespace hipjpeg
