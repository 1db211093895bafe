// ThreadSanitizer harness for ForkJoinPool (csrc/thread_pool.h): several threads call parallel_for on ONE pool at once -- what the
// plugin does when its completion thread decodes fallback images in resolve() while the caller's thread plans the next piece.  Every
// call must run each of its indices exactly once, return only after its own lambdas have finished, and propagate its own exception.
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <thread>
#include <vector>

#include "thread_pool.h"

int main()
{
    hipjpeg::ForkJoinPool pool(4);
    std::atomic<int> bad{0};
    auto caller = [&](int id) {
        for (int round = 0; round < 400; round++) {
            const int n = 2 + (round * 7 + id * 3) % 37;
            std::vector<std::atomic<int>> hits(n);
            for (auto& h : hits) h.store(0);
            std::vector<int> scratch(n, 0);  // dies with this iteration: a helper still running the lambda would touch freed memory
            bool threw = false;
            try {
                pool.parallel_for(n, [&](int i, int) {
                    hits[i].fetch_add(1);
                    scratch[i] = i * id;
                    if (round % 50 == 49 && i == n / 2) throw std::runtime_error("boom");
                });
            } catch (const std::runtime_error&) {
                threw = true;
            }
            if (round % 50 == 49) {
                if (!threw) bad++;
                for (int i = 0; i < n; i++)
                    if (hits[i].load() > 1) bad++;
            } else {
                if (threw) bad++;
                for (int i = 0; i < n; i++)
                    if (hits[i].load() != 1 || scratch[i] != i * id) bad++;
            }
        }
    };
    std::vector<std::thread> th;
    for (int id = 1; id <= 3; id++) th.emplace_back(caller, id);
    for (auto& t : th) t.join();
    std::printf("pool_race: %d violations\n", bad.load());
    return bad.load() ? 1 : 0;
}
