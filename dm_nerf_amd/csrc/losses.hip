// losses.hip -- the scalar tail of the training step (train_dmsr.py:33-61) for BOTH levels in one launch each way.
//
// The reference forms   loss = sum over (fine, coarse) of  img2mse(rgb, target) + ins_criterion(ins, labels) [+ ins_penalizer(...)]
// out of a few dozen small tensor operations; on this path every one of them is a kernel of a few microseconds, and at the
// per-rank shard of an 8-way split (384 rays) those launches -- not the arithmetic -- were a quarter of the step (profiles/r04).
// Here the tail is two kernels:
//   loss_tail_fwd_kernel   mean squared error of both levels (evaluator.py:11), the penalizer's scalar tail of both levels
//                          (penalizer.py:43-55, from the four batch sums the partial-sum kernel produced), and the total in the
//                          order the training loop adds the terms (fine: mse, criterion, penalizer; then coarse), all f32
//   loss_tail_bwd_kernel   d loss / d rgb of both levels = (g / n) * (2 * (rgb - target)) -- the products autograd forms for
//                          mean((x - y) ** 2) -- and the upstream factors the criterion / penalizer backward kernels read
//                          (gout4 = {g, 0, 0, 0}; scales = inv * g) so that no scalar tensor arithmetic is left on the stream
// The object-code loss itself (csrc/criterion.hip, two levels per launch) and the penalizer's per-ray kernels
// (csrc/render_kernels.hip) are the ones the drop-in functions use.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dmnerf_hip.h"
#include "common.h"

namespace {

struct TailArgs {
    const float* rgb[2];
    const float* target;
    int64_t N;                   // rows of rgb / target ([N, 3])
    const float* crit_out4[2];   // nullable: ins_criterion outputs (out4[0] = the loss term)
    const double* pen_sums4[2];  // nullable: the penalizer's four batch sums per level
    int C;                       // object channels incl. the "empty" one (penalizer normaliser)
    float* terms8;               // mse_a, crit_a, pen_a, mse_b, crit_b, pen_b, total, 0
    float* pen_inv4;             // 1 / (C max(sum m_b, 1e-8)), 1 / max(sum m_m, 1e-8) per level
};

__global__ __launch_bounds__(1024) void loss_tail_fwd_kernel(const TailArgs a) {
    __shared__ double red[1024];
    __shared__ float s_mse[2];
    const int64_t n = a.N * 3;
    for (int lvl = 0; lvl < 2; ++lvl) {
        const float* __restrict__ x = a.rgb[lvl];
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < n; i += 1024) {
            const float d = x[i] - a.target[i];
            s += (double)(d * d);                                  // f32 square like (x - y) ** 2, accumulated in double
        }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 512; o >= 1; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) s_mse[lvl] = (float)(red[0] / (double)n);
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    float crit2[2] = {0.f, 0.f}, pen2[2] = {0.f, 0.f};
    for (int lvl = 0; lvl < 2; ++lvl) {
        float pen = 0.f;
        if (a.pen_sums4[lvl]) {                                    // penalizer.py:43-55 (same arithmetic as penalizer_finish_kernel)
            const double* s4 = a.pen_sums4[lvl];
            const double nb = s4[1] > 1e-8 ? s4[1] : 1e-8, nm = s4[3] > 1e-8 ? s4[3] : 1e-8;
            pen = (float)(s4[0] / ((double)a.C * nb) + s4[2] / nm);
            a.pen_inv4[2 * lvl + 0] = (float)(1.0 / ((double)a.C * nb));
            a.pen_inv4[2 * lvl + 1] = (float)(1.0 / nm);
        } else {
            a.pen_inv4[2 * lvl + 0] = 0.f;
            a.pen_inv4[2 * lvl + 1] = 0.f;
        }
        const float crit = a.crit_out4[lvl] ? a.crit_out4[lvl][0] : 0.f;
        a.terms8[3 * lvl + 0] = s_mse[lvl];
        a.terms8[3 * lvl + 1] = crit;
        a.terms8[3 * lvl + 2] = pen;
        crit2[lvl] = crit;
        pen2[lvl] = pen;
    }
    // The reference's association order (train_dmsr.py:47-58; level a = fine, b = coarse):
    //   ins_loss = ins_fine + ins_coarse; rgb_loss = rgb_fine + rgb_coarse; total = ins_loss + rgb_loss;
    //   [penalize] total = total + (emptiness_fine + emptiness_coarse)
    // (each term is this library's f32 value -- the squared error is summed in double here, an f32 tree in ATen -- so the total is
    // the reference's up to the rounding of the terms, not bit for bit)
    float total = s_mse[0] + s_mse[1];
    if (a.crit_out4[0] || a.crit_out4[1]) total = (crit2[0] + crit2[1]) + total;
    if (a.pen_sums4[0] || a.pen_sums4[1]) total = total + (pen2[0] + pen2[1]);
    a.terms8[6] = total;
    a.terms8[7] = 0.f;
}

struct TailBwdArgs {
    const float* rgb[2];
    const float* target;
    int64_t N;
    const float* g_total;        // device scalar: d L / d total
    const float* pen_inv4;
    float* d_rgb[2];
    float* gout8;                // {g, 0, 0, 0} per level: upstream of the criterion's four outputs
    float* pen_scales4;          // inv * g per level
    double* g_part8;             // nullable: {scale_b, 0, scale_m, 0} per level, the row dmnerf_composite_pen_bwd reads
};

__global__ __launch_bounds__(256) void loss_tail_bwd_kernel(const TailBwdArgs a) {
    const float g = a.g_total[0];
    const int64_t n = a.N * 3;
    const float gn = g / (float)n;                                 // MeanBackward: grad / numel
    const int lvl = blockIdx.y;
    const float* __restrict__ x = a.rgb[lvl];
    float* __restrict__ dx = a.d_rgb[lvl];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = x[i] - a.target[i];
        dx[i] = gn * (2.f * d);                                    // PowBackward0: grad * (2 * (x - y))
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        a.gout8[4 * lvl + threadIdx.x] = threadIdx.x == 0 ? g : 0.f;
        if (threadIdx.x < 2) {
            const float sc = a.pen_inv4[2 * lvl + threadIdx.x] * g;
            a.pen_scales4[2 * lvl + threadIdx.x] = sc;
            if (a.g_part8) {
                a.g_part8[4 * lvl + 2 * threadIdx.x] = (double)sc;
                a.g_part8[4 * lvl + 2 * threadIdx.x + 1] = 0.0;
            }
        }
    }
}

}  // namespace

extern "C" int dmnerf_loss_tail_fwd(const float* d_rgb_a, const float* d_rgb_b, const float* d_target, int64_t N,
                                    const float* d_crit_out4_a, const float* d_crit_out4_b, const double* d_pen_sums4_a,
                                    const double* d_pen_sums4_b, int C, float* d_terms8, float* d_pen_inv4, void* stream) {
    if (N < 1 || C < 1) return dmn_fail(DMNERF_E_ARG, "loss_tail_fwd: bad N=%lld C=%d", (long long)N, C);
    if (!d_rgb_a || !d_rgb_b || !d_target || !d_terms8 || !d_pen_inv4) return dmn_fail(DMNERF_E_ARG, "loss_tail_fwd: null pointer");
    TailArgs a{};
    a.rgb[0] = d_rgb_a; a.rgb[1] = d_rgb_b; a.target = d_target; a.N = N;
    a.crit_out4[0] = d_crit_out4_a; a.crit_out4[1] = d_crit_out4_b;
    a.pen_sums4[0] = d_pen_sums4_a; a.pen_sums4[1] = d_pen_sums4_b;
    a.C = C; a.terms8 = d_terms8; a.pen_inv4 = d_pen_inv4;
    hipLaunchKernelGGL(loss_tail_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    return dmn_check_launch("loss_tail_fwd");
}

extern "C" int dmnerf_loss_tail_bwd(const float* d_rgb_a, const float* d_rgb_b, const float* d_target, int64_t N,
                                    const float* d_g_total, const float* d_pen_inv4, float* d_grad_rgb_a, float* d_grad_rgb_b,
                                    float* d_gout8, float* d_pen_scales4, double* d_g_partials8, void* stream) {
    if (N < 1) return dmn_fail(DMNERF_E_ARG, "loss_tail_bwd: bad N=%lld", (long long)N);
    if (!d_rgb_a || !d_rgb_b || !d_target || !d_g_total || !d_pen_inv4 || !d_grad_rgb_a || !d_grad_rgb_b || !d_gout8 || !d_pen_scales4)
        return dmn_fail(DMNERF_E_ARG, "loss_tail_bwd: null pointer");
    TailBwdArgs a{};
    a.rgb[0] = d_rgb_a; a.rgb[1] = d_rgb_b; a.target = d_target; a.N = N; a.g_total = d_g_total; a.pen_inv4 = d_pen_inv4;
    a.d_rgb[0] = d_grad_rgb_a; a.d_rgb[1] = d_grad_rgb_b; a.gout8 = d_gout8; a.pen_scales4 = d_pen_scales4; a.g_part8 = d_g_partials8;
    const int64_t n = N * 3;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(loss_tail_bwd_kernel, dim3(blocks, 2), dim3(256), 0, (hipStream_t)stream, a);
    return dmn_check_launch("loss_tail_bwd");
}
