// thread_pool.h -- minimal fork/join worker pool for the host entropy stage of the batched C API.
// (The nvImageCodec plugin path does not use it: there the framework's executor supplies the threads,
//  nvimgcodecExecutorDesc_t, reference include/nvimgcodec.h:800-828.)
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <exception>
#include <thread>
#include <vector>

namespace hipjpeg {

// CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (containers often see every core
// of the host in hardware_concurrency() while being allowed a fraction of them).
inline int usable_cpus()
{
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm' && period > 0) {
            quota = atoll(q);
            const int cap = (int)((quota + period - 1) / period);
            if (cap > 0 && cap < n) n = cap;
        }
        fclose(f);
    }
    return n > 0 ? n : 1;
}

class ForkJoinPool {
public:
    explicit ForkJoinPool(int num_threads)
    {
        if (num_threads <= 0) num_threads = usable_cpus();
        nthreads_ = num_threads;
        // the calling thread participates, so spawn nthreads-1 helpers
        for (int t = 1; t < nthreads_; t++) workers_.emplace_back([this, t] { worker_loop(t); });
    }
    ~ForkJoinPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            generation_++;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    int num_threads() const { return nthreads_; }

    // Runs fn(index, thread_id) for index in [0, n); indices are handed out dynamically.  Blocks until done.
    // An exception thrown by fn on any thread stops the hand-out of further indices and is rethrown here, on the calling
    // thread, once every helper has come back (a throw inside a helper thread would otherwise end the process).
    // Two threads may call at once (the plugin's completion thread decodes fallback images in resolve() while the caller's thread
    // plans the next piece on the same pool): the helpers serve one job at a time, so the caller that does not get them runs its
    // indices itself, with the thread id it would have had as the pool's calling thread -- callers that index per-thread scratch
    // by that id must not share such scratch between concurrent calls (none does: resolve() and plan() keep theirs on the stack).
    void parallel_for(int n, const std::function<void(int, int)>& fn)
    {
        if (n <= 0) return;
        if (nthreads_ == 1 || n == 1) {
            for (int i = 0; i < n; i++) fn(i, 0);
            return;
        }
        std::unique_lock<std::mutex> owner(caller_m_, std::try_to_lock);
        if (!owner.owns_lock()) {
            for (int i = 0; i < n; i++) fn(i, 0);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            job_n_ = n;
            next_.store(0);
            pending_ = nthreads_ - 1;
            failure_ = nullptr;
            generation_++;
        }
        cv_.notify_all();
        run_indices(fn, n, 0);
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
        if (failure_) {
            std::exception_ptr e = failure_;
            failure_ = nullptr;
            lk.unlock();
            std::rethrow_exception(e);
        }
    }

private:
    void run_indices(const std::function<void(int, int)>& fn, int n, int tid)
    {
        for (;;) {
            int i = next_.fetch_add(1);
            if (i >= n) break;
            try {
                fn(i, tid);
            } catch (...) {
                std::lock_guard<std::mutex> lk(m_);
                if (!failure_) failure_ = std::current_exception();
                next_.store(n);  // nobody starts another index
            }
        }
    }
    void worker_loop(int tid)
    {
        unsigned seen = 0;
        for (;;) {
            const std::function<void(int, int)>* job;
            int n;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                job = job_;
                n = job_n_;
            }
            if (job) run_indices(*job, n, tid);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_cv_.notify_all();
            }
        }
    }

    int nthreads_ = 1;
    std::vector<std::thread> workers_;
    std::mutex m_, caller_m_;  // caller_m_: held by the one parallel_for call that has the helpers
    std::condition_variable cv_, done_cv_;
    const std::function<void(int, int)>* job_ = nullptr;
    int job_n_ = 0, pending_ = 0;
    unsigned generation_ = 0;
    bool stop_ = false;
    std::atomic<int> next_{0};
    std::exception_ptr failure_;
};

}  // namespace hipjpeg
