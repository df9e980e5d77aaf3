// optim.hip -- the optimizer update and the weight re-packing of a training step as TWO launches (gfx950).  EXTENSION: the
// reference steps torch.optim.Adam over list(coarse.parameters()) + list(fine.parameters()) (train_dmsr.py:124-125, :62-64),
// which is what the drop-in path keeps doing; at the 384-ray shard of an 8-way split those ~9 multi-tensor launches plus the
// 8 pack / cat / head-product launches that follow are 0.17 ms of a 3.4 ms step (docs/EXPERIMENTS.md, round 5).
//
//   adam_kernel     one pass over the FLAT parameter / gradient / moment vectors of all models (the gradient is the GradArena
//                   the weight-gradient kernels wrote into): the update of torch.optim.Adam's formula, element for element
//                   (torch/optim/adam.py::_multi_tensor_adam, amsgrad = False, weight_decay = 0, maximize = False):
//                       m <- lerp(m, g, 1 - b1);  v <- v b2;  v <- v + (1 - b2) (g g)
//                       step_size = lr / (1 - b1^t);  denom = sqrt(v) / sqrt(1 - b2^t) + eps;  p <- p + (-step_size) (m / denom)
//                   the same f32 operations in the same order (the three multiply-adds fused, as in ATen's device kernels), the
//                   scalar factors formed in double and rounded once, as the Python side of the reference optimizer does.
//                   HBM-bound: 28 B per parameter (read g p m v, write p m v).
//   repack_kernel   per model, from the UPDATED flat parameters: a fresh copy of the flat vector (what a pending backward's head
//                   kernels must keep seeing), the forward blob (gather, = dmnerf_pack_weights) and the W^T blob (gather from
//                   [parameters | F] with F = A W_rf formed inline by the same f32 fmaf chain, k ascending, as
//                   head_product_kernel) -- bit-identical to dmnerf_head_product + 2 x dmnerf_pack_weights.
// The step counter lives on the device (graph-capturable): every workgroup reads t = state[0] + 1 when it starts; the LAST
// workgroup to finish (atomic ticket in state[1]) stores the new count and resets the ticket.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "layout.h"
#include "params.h"

using namespace dmn;

namespace {

struct AdamArgs {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    const float* d_lr;      // nullable: device scalar learning rate (f32), else `lr`
    double lr, beta1, beta2, eps;
    long long* state;       // [0] step count t (number of updates made), [1] ticket of finished workgroups
    int vec4;               // all four vectors 16-byte aligned
};
typedef float4 f4;

struct AdamScalars {
    float neg_step, bc2_sqrt, w1, w2, b2, eps;
    long long t;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamScalars& k) {
    // (the three multiply-adds are FUSED, as the device compiler contracts ATen's `self + weight * diff`, `a + alpha * (b * c)`
    // and `a + alpha * (b / c)` in the foreach kernels; this file is built with -ffp-contract=off, so it is spelled out)
    m = fmaf(k.w1, g - m, m);                                       // _foreach_lerp_(exp_avgs, grads, 1 - beta1): weight < 0.5 form
    v = v * k.b2;                                                   // _foreach_mul_(exp_avg_sqs, beta2)
    v = fmaf(k.w2, g * g, v);                                       // _foreach_addcmul_(exp_avg_sqs, grads, grads, 1 - beta2)
    float d = sqrtf(v);                                             // _foreach_sqrt
    d = d / k.bc2_sqrt;                                             // _foreach_div_(., bias_correction2_sqrt)
    d = d + k.eps;                                                  // _foreach_add_(., eps)
    p = fmaf(k.neg_step, m / d, p);                                 // _foreach_addcdiv_(params, exp_avgs, ., step_size)
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
    __shared__ AdamScalars sk;
    if (threadIdx.x == 0) {                                         // the scalar factors once per workgroup (two double pow)
        // (agent-scope atomic load: the step count is written by ONE workgroup of the previous launch on whatever XCD it ran; this
        // read must not be served from a stale line of this XCD's L2 whatever the cache policy of the allocation -- it does not
        // lean on the kernel-boundary write-back / invalidate.  One access per workgroup.)
        const long long t = __hip_atomic_load(&a.state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        const double lr = a.d_lr ? (double)a.d_lr[0] : a.lr;
        const double bc1 = 1.0 - pow(a.beta1, (double)t), bc2 = 1.0 - pow(a.beta2, (double)t);
        sk.neg_step = (float)(lr / bc1 * -1.0);                     // adam.py: step_size = (lr / bias_correction1) * -1
        sk.bc2_sqrt = (float)sqrt(bc2);                             //          bias_correction2_sqrt = bias_correction2 ** 0.5
        sk.w1 = (float)(1.0 - a.beta1); sk.w2 = (float)(1.0 - a.beta2); sk.b2 = (float)a.beta2; sk.eps = (float)a.eps;
        sk.t = t;
    }
    __syncthreads();
    const AdamScalars k = sk;
    const int64_t n4 = a.vec4 ? a.n / 4 : 0;                        // 16-byte path when all four vectors are 16-byte aligned
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f4 g = reinterpret_cast<const f4*>(a.g)[i];
        f4 p = reinterpret_cast<f4*>(a.p)[i], m = reinterpret_cast<f4*>(a.m)[i], v = reinterpret_cast<f4*>(a.v)[i];
        adam_one(p.x, g.x, m.x, v.x, k); adam_one(p.y, g.y, m.y, v.y, k); adam_one(p.z, g.z, m.z, v.z, k); adam_one(p.w, g.w, m.w, v.w, k);
        reinterpret_cast<f4*>(a.p)[i] = p; reinterpret_cast<f4*>(a.m)[i] = m; reinterpret_cast<f4*>(a.v)[i] = v;
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        float p = a.p[i], m = a.m[i], v = a.v[i];
        adam_one(p, a.g[i], m, v, k);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (no release fence: the ticket only orders the READS of state[0], made at the top of every workgroup, before the one write
        // below; an agent-scope release here would write back this XCD's L2 once per workgroup -- measured: 35 us instead of 12 for
        // the pass.  This workgroup's own read has RETURNED by now: its value went through `sk` in LDS and the barrier above.
        // All three accesses to `state` are agent-scope atomics, so none of them is a plain load / store racing across XCD L2s.)
        const unsigned long long done = __hip_atomic_fetch_add((unsigned long long*)&a.state[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
        if (done == gridDim.x) {                                    // every workgroup has read state[0] (it read it before it finished)
            __hip_atomic_store(&a.state[1], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.state[0], k.t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct RepackModel {
    const float* flat;          // updated parameters of this model (a slice of the optimizer's flat vector)
    float* flat_copy;           // nullable
    const int32_t* idx;         // forward blob gather index
    float* blob;
    const int32_t* idx_t;       // W^T blob gather index over [parameters | F]
    float* blob_t;
    const int32_t* f_pos;       // nullable: position in blob_t of every F element (inverse of idx_t on its F entries)
    int64_t n_blob, n_blob_t;
    Params P;
};
struct RepackArgs {
    RepackModel m[DMNERF_REPACK_MAX_MODELS];
};

// F[i][j] = sum_k A[i][k] W_rf[k][j], f32 fmaf chain over k ascending (= heads.hip::head_product_kernel)
__device__ __forceinline__ float head_f(const float* __restrict__ flat, const Params& P, int i, int j) {
    const float* __restrict__ arow = flat + P.rgb_hidden.w_off + (int64_t)i * P.rgb_hidden.in;
    const float* __restrict__ wcol = flat + P.rgb_feature.w_off + j;
    float acc = 0.f;
    for (int q = 0; q < W; ++q) acc = fmaf(arow[q], wcol[(int64_t)q * W], acc);
    return acc;
}

__global__ __launch_bounds__(256) void repack_kernel(const RepackArgs a) {
    const RepackModel& M = a.m[blockIdx.y];
    const int64_t n_param = M.P.total;
    const int64_t n_copy = M.flat_copy ? n_param : 0;
    const int64_t n_f = M.f_pos ? HEAD_F_FLOATS : 0;               // F in its NATURAL order (consecutive lanes: consecutive columns j of
    const int64_t n_all = n_f + n_copy + M.n_blob + M.n_blob_t;    // W_rf, coalesced; one row of A, broadcast), scattered to its blob slots
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_all; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < n_f) {                                              // (first: these threads run the longest)
            M.blob_t[M.f_pos[e]] = head_f(M.flat, M.P, (int)(e / W), (int)(e % W));
        } else if (e < n_f + n_copy) {
            const int64_t k = e - n_f;
            M.flat_copy[k] = M.flat[k];
        } else if (e < n_f + n_copy + M.n_blob) {
            const int64_t k = e - n_f - n_copy;
            const int32_t s = M.idx[k];
            M.blob[k] = s >= 0 ? M.flat[s] : 0.f;
        } else {
            const int64_t k = e - n_f - n_copy - M.n_blob;
            const int32_t s = M.idx_t[k];
            if (s >= n_param) {
                if (!M.f_pos) M.blob_t[k] = head_f(M.flat, M.P, (int)((s - n_param) / W), (int)((s - n_param) % W));   // (gather form: slow, uncoalesced)
            } else {
                M.blob_t[k] = s >= 0 ? M.flat[s] : 0.f;
            }
        }
    }
}

}  // namespace

extern "C" int dmnerf_adam_step(float* d_params, const float* d_grads, float* d_exp_avg, float* d_exp_avg_sq, int64_t n,
                                double lr, const float* d_lr, double beta1, double beta2, double eps, int64_t* d_state2, void* stream) {
    if (n < 0) return dmn_fail(DMNERF_E_ARG, "adam_step: n = %lld", (long long)n);
    if (n == 0) return DMNERF_OK;
    if (!d_params || !d_grads || !d_exp_avg || !d_exp_avg_sq || !d_state2) return dmn_fail(DMNERF_E_ARG, "adam_step: null pointer");
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
        return dmn_fail(DMNERF_E_ARG, "adam_step: beta1 %g / beta2 %g / eps %g out of range", beta1, beta2, eps);
    if (1.0 - beta1 >= 0.5) return dmn_fail(DMNERF_E_ARG, "adam_step: beta1 %g <= 0.5 (the other branch of lerp) is not implemented", beta1);
    const uintptr_t al = (uintptr_t)d_params | (uintptr_t)d_grads | (uintptr_t)d_exp_avg | (uintptr_t)d_exp_avg_sq;
    AdamArgs a{d_params, d_grads, d_exp_avg, d_exp_avg_sq, n, d_lr, lr, beta1, beta2, eps, (long long*)d_state2, (al & 15) == 0 ? 1 : 0};
    const int64_t blocks = (n + 256 * 8 - 1) / (256 * 8);          // two 16-byte quads per thread
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream, a);
    return dmn_check_launch("adam_step");
}

extern "C" int dmnerf_repack_train(const dmnerf_repack_model* models, int n_models, void* stream) {
    if (!models || n_models < 1 || n_models > DMNERF_REPACK_MAX_MODELS)
        return dmn_fail(DMNERF_E_ARG, "repack_train: 1..%d models, got %d", DMNERF_REPACK_MAX_MODELS, n_models);
    RepackArgs a{};
    int64_t most = 0;
    for (int i = 0; i < n_models; ++i) {
        const dmnerf_repack_model& s = models[i];
        if (s.ins_num < 1 || s.ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "repack_train: ins_num %d unsupported", s.ins_num);
        if (!s.d_params_flat || !s.d_idx || !s.d_blob || !s.d_idx_t || !s.d_blob_t) return dmn_fail(DMNERF_E_ARG, "repack_train: null pointer");
        if (s.d_flat_copy == s.d_params_flat) return dmn_fail(DMNERF_E_ARG, "repack_train: d_flat_copy aliases d_params_flat");
        RepackModel& m = a.m[i];
        m.flat = s.d_params_flat; m.flat_copy = s.d_flat_copy; m.idx = s.d_idx; m.blob = s.d_blob; m.idx_t = s.d_idx_t; m.blob_t = s.d_blob_t;
        m.f_pos = s.d_f_pos;
        m.n_blob = dmnerf_blob_floats(s.ins_num);
        m.n_blob_t = dmnerf_blob_t_floats(s.ins_num);
        m.P = make_params(s.ins_num);
        const int64_t n_all = m.P.total + m.n_blob + m.n_blob_t + HEAD_F_FLOATS;
        if (n_all > most) most = n_all;
    }
    const int64_t blocks = (most + 255) / 256;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096), (unsigned)n_models), dim3(256), 0, (hipStream_t)stream, a);
    return dmn_check_launch("repack_train");
}
