// diagnostics.cpp -- see diagnostics.h
#include "diagnostics.h"

#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

namespace hipjpeg {

namespace {

using PushFn = int (*)(const char*);
using PopFn = int (*)();

struct Roctx {
    PushFn push = nullptr;
    PopFn pop = nullptr;
    Roctx()
    {
        if (getenv("HIPJPEG_NO_ROCTX")) return;
        for (const char* lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<PushFn>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<PopFn>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};

const Roctx& roctx()
{
    static const Roctx r;
    return r;
}

std::atomic<int> g_fault_armed{0};
std::mutex g_fault_mutex;
std::string g_fault_site;
int g_fault_countdown = 0;

}  // namespace

void range_push(const char* name)
{
    const Roctx& r = roctx();
    if (r.push) r.push(name);
}

void range_pop()
{
    const Roctx& r = roctx();
    if (r.pop) r.pop();
}

void fault_point(const char* site)
{
    if (!g_fault_armed.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_fault_mutex);
    if (g_fault_site != site) return;
    if (--g_fault_countdown > 0) return;
    g_fault_armed.store(0);
    g_fault_site.clear();
    throw std::runtime_error(std::string("injected fault at ") + site);
}

void set_fault(const char* site, int countdown)
{
    std::lock_guard<std::mutex> lk(g_fault_mutex);
    if (!site || !*site || countdown < 1) {
        g_fault_site.clear();
        g_fault_armed.store(0);
        return;
    }
    g_fault_site = site;
    g_fault_countdown = countdown;
    g_fault_armed.store(1);
}

}  // namespace hipjpeg
