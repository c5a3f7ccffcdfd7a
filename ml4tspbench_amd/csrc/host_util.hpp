// ml4tspbench_amd/csrc/host_util.hpp -- error reporting and device guard shared by the C-ABI sources
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "tsp2opt.h"

namespace t2o {

inline thread_local std::string g_err;

inline int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(TSP2OPT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace t2o
