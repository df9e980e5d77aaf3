// diag.hip -- run-time honesty of the opt-in f16x2 mode (include/dmnerf_hip.h: dmnerf_f16x2_range_flags).
//
// The split-f16 kernels convert every activation (forward) and every scaled gradient (backward) with v_cvt_pkrtz_f16_f32, which
// SATURATES at 65 504 instead of overflowing (split_f16.h): a network whose activations leave the f16 range keeps producing
// finite, wrong values.  The hand-scheduled kernels have no free issue slot for a comparison per converted pair, but every
// value they convert is also what the training forward SAVES (f32 rows of the SaveLayout workspace, csrc/layout.h) and what the
// data-gradient kernel writes (same layout): one pass over those rows finds exactly the operands that saturated.  Opt-in
// (DMNERF_CHECK_F16=1 / args.check_f16): ~2 ms per 4096-ray launch; nothing runs otherwise.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "layout.h"

namespace {

__global__ __launch_bounds__(256) void range_flags_kernel(const float* __restrict__ x, int64_t n, float limit, int bit,
                                                          int32_t* __restrict__ flags) {
    bool bad = false;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            bad = bad || !(fabsf(v.x) < limit) || !(fabsf(v.y) < limit) || !(fabsf(v.z) < limit) || !(fabsf(v.w) < limit);   // (NaN counts)
        } else {
            for (int64_t j = i; j < n; ++j) bad = bad || !(fabsf(x[j]) < limit);
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, bit);
}

}  // namespace

extern "C" int dmnerf_f16x2_range_flags(const float* d_workspace, int64_t M, int gradients, int32_t* d_flags, void* stream) {
    if (M < 0 || !d_flags || (M > 0 && !d_workspace)) return dmn_fail(DMNERF_E_ARG, "f16x2_range_flags: bad argument");
    if (M == 0) return DMNERF_OK;
    const dmn::SaveLayout s = dmn::make_save_layout(M);
    // forward workspace: pe | de | h_0..h_7 | g1 | g2 (everything the next layer converts); gradient workspace: h | g rows
    const int64_t lo = gradients ? s.h : s.pe, hi = s.bits;
    const int64_t n = hi - lo;
    const unsigned blocks = (unsigned)((n / 4 + 255) / 256 < 8192 ? (n / 4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(range_flags_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, (hipStream_t)stream, d_workspace + lo, n, 65504.0f,
                       gradients ? DMNERF_F16_GRAD_SATURATED : DMNERF_F16_ACT_SATURATED, d_flags);
    return dmn_check_launch("f16x2_range_flags");
}
