// tests/native/tsan_driver.cpp -- host-side thread-safety check of libtsp2opt_hip's C ABI (CPU only).
// Built with ThreadSanitizer on the HOST code of the library sources (tests/test_tsan_host.py); no GPU
// is touched: only the entry points that make no HIP call (planners, debug knobs, argument checks,
// error strings).  Every thread sets its own knobs and must keep seeing exactly their effect while the
// other threads set different ones (the knobs are per calling thread), and TSan must stay silent.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "tsp2opt.h"

static std::atomic<int> g_fail{0};

static void worker(int id, int rounds)
{
    const int nt = 64 << (id % 4);                 // 64, 128, 256, 512 threads per workgroup
    const int mode = id % 2 ? 0 : 4;               // cluster choice off / forced to 4 workgroups per tour
    for (int r = 0; r < rounds; ++r) {
        tsp2opt_debug_tune(0, nt);
        tsp2opt_debug_cluster_mode(mode);
        tsp2opt_debug_tune(4, 5 + id % 8);
        tsp2opt_debug_tune(6, id & 1);
        tsp2opt_profile_enable(id & 1);
        tsp2opt_debug_sweep_only(0);
        int32_t out7[7] = {0}, out12[12] = {0};
        const int n = 200 + 37 * ((r + id) % 20);
        if (tsp2opt_plan(n, TSP2OPT_DTYPE_F64, -1, out7) != 0 || out7[0] != nt) g_fail.fetch_add(1);
        if (tsp2opt_cluster_plan(800, TSP2OPT_DTYPE_F32, 3, 256, out12) != 0) g_fail.fetch_add(1);
        if ((mode == 0) != (out12[0] == 0)) g_fail.fetch_add(1);
        if (tsp2opt_workspace_bytes(1000 + r, n, 1) <= 0) g_fail.fetch_add(1);
        // error path: the message is the calling thread's own
        if (id & 1) {
            if (tsp2opt_solve(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 0, 1, 1, 10, 0.0, 1, 0,
                              nullptr, 0, nullptr) != TSP2OPT_E_ARG || !strstr(tsp2opt_last_error(), "n < 1"))
                g_fail.fetch_add(1);
        } else {
            if (tsp2opt_solve_noalloc((void*)8, (int64_t*)8, nullptr, nullptr, nullptr, nullptr, 4, 10, 1, 1, 10,
                                      0.0, 7, 0, nullptr) != TSP2OPT_E_ARG || !strstr(tsp2opt_last_error(), "dtype"))
                g_fail.fetch_add(1);
        }
        double ms = 0.0;
        if (tsp2opt_profile_history_ms(id % 4, 0, &ms, nullptr) == 0) g_fail.fetch_add(1);   // nothing profiled
    }
}

int main()
{
    std::vector<std::thread> th;
    for (int id = 0; id < 4; ++id) th.emplace_back(worker, id, 400);
    for (auto& t : th) t.join();
    std::printf("tsan_driver: %d failures\n", g_fail.load());
    return g_fail.load() ? 1 : 0;
}
