// Dev tool (GPU box): what hipPointerGetAttributes costs per call on pageable and on pinned host pointers -- the zero-copy input path asks
// it once per image (decoder_core.cpp plan).  build: hipcc -O2 tools/pointer_query_cost.cpp -o /tmp/pqc
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main()
{
    hipFree(nullptr);
    const int n = 4096;
    std::vector<void*> pageable(n), pinned(n);
    for (int i = 0; i < n; i++) {
        pageable[i] = malloc(500000);
        if (i < 256) (void)hipHostMalloc(&pinned[i], 500000);
    }
    for (int round = 0; round < 2; round++) {
        auto t0 = std::chrono::steady_clock::now();
        int hits = 0;
        for (int i = 0; i < n; i++) {
            hipPointerAttribute_t a;
            if (hipPointerGetAttributes(&a, (char*)pageable[i] + 600) == hipSuccess) hits++;
            else (void)hipGetLastError();
        }
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < 256; i++) {
            hipPointerAttribute_t a;
            if (hipPointerGetAttributes(&a, (char*)pinned[i] + 600) == hipSuccess && a.type == hipMemoryTypeHost) hits++;
        }
        auto t2 = std::chrono::steady_clock::now();
        printf("pageable: %.2f us per query, pinned: %.2f us per query (%d hits)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / n,
               std::chrono::duration<double, std::micro>(t2 - t1).count() / 256, hits);
    }
    return 0;
}
