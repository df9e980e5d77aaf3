// common.h -- error plumbing shared by the host-side translation units.
#pragma once
#include <atomic>
#include <cstdarg>
#include <cstdio>

#include <hip/hip_runtime.h>

// Records a printf-style message for dmnerf_last_error() and returns `code`.
int dmn_fail(int code, const char* fmt, ...);
// After a kernel launch: 0, or DMNERF_E_LAUNCH with hipGetErrorString recorded.
int dmn_check_launch(const char* what);
// A HIP runtime call that FAILED with `e` (hipFuncSetAttribute, hipMemsetAsync ...): always an error return.
// (dmn_check_launch() is the wrong tool there: it reports hipGetLastError(), which may have nothing to say, and the
// caller would return DMNERF_OK without having launched anything.)
int dmn_fail_hip(hipError_t e, const char* what);

// hipFuncSetAttribute (large dynamic LDS) is per function AND per device: one bit per device ordinal in a per-call-site
// mask, so a process that drives several GPUs configures each of them, and two host threads racing here both succeed.
struct DmnOncePerDevice {
    std::atomic<unsigned long long> mask{0};
    template <class F>
    hipError_t run(F&& f) {
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
        const unsigned long long bit = 1ull << (dev & 63);
        if (mask.load(std::memory_order_acquire) & bit) return hipSuccess;
        if (hipError_t e = f(); e != hipSuccess) return e;
        mask.fetch_or(bit, std::memory_order_release);
        return hipSuccess;
    }
};
