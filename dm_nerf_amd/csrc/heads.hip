// heads.hip -- the two activation-free "feature" linears of DM_NeRF in the BACKWARD pass (gfx950).
//
// rgb_feature_linear and ins_feature_linear have no activation (networks/dm_nerf.py:89,96): with h = h_7,
//     f = W_rf h + b_rf,    g1 = relu(A f + B dirs + b_rh)        (A | B = rgb_feature_linears.0.weight[:, :256 | 256:])
//     q = W_if h.detach() + b_if,    g2 = relu(W_ih q + b_ih)
// so everything the backward needs from f and q is linear in h and can be re-associated (exact algebra, f32 rounding
// of a different summation order -- the class of difference the split-K weight gradient has anyway):
//     d h (rgb branch) = W_rf^T A^T dg1 = F^T dg1,   F = A W_rf  [128 x 256]      (one GEMM in the dgrad kernel)
//     d A   = sum dg1 f^T = G W_rf^T + s1 b_rf^T,    G = sum dg1 h^T [128 x 256],  s1 = sum dg1 (= d b_rh)
//     d W_rf = sum (A^T dg1) h^T = A^T G,            d b_rf = A^T s1
//     d W_ih = Q W_if^T + s2 b_if^T,                 Q = sum dg2 h^T,              s2 = sum dg2 (= d b_ih)
//     d W_if = W_ih^T Q,                             d b_if = W_ih^T s2
// The training forward therefore does not save f and q (512 of 2906 rows per sample), the dgrad kernel runs 31 instead
// of 37 weight quarters and writes neither d f nor d q, and the weight-gradient kernel forms G and Q (2 x 128 x 256
// outputs over the samples) instead of four products with 256 x 256 + 256 x 256 + 128 x 256 + 128 x 256 outputs:
// 131 072 fewer MACs per sample in each of the two backward kernels.  What is left for this file is O(parameters):
//   head_product_kernel   F = A W_rf, once per weight update (behind the flat parameter vector; the W^T blob gathers it)
//   head_unfuse_kernel    the six small products above, once per backward, into the flat gradient vector
// Both are plain f32 fmaf chains over k ascending (<= 256 terms): no vendor BLAS.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "layout.h"
#include "params.h"

using namespace dmn;

namespace {

__global__ void head_product_kernel(const float* __restrict__ flat, const Params P, float* __restrict__ F) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;             // F[i][j], i < 128, j < 256
    if (e >= HW * W) return;
    const int i = e / W, j = e % W;
    const float* __restrict__ a = flat + P.rgb_hidden.w_off + (int64_t)i * P.rgb_hidden.in;   // A[i][:]
    const float* __restrict__ w = flat + P.rgb_feature.w_off + j;                              // W_rf[:][j]
    float s = 0.f;
    for (int k = 0; k < W; ++k) s = fmaf(a[k], w[(int64_t)k * W], s);
    F[e] = s;
}

// Inference-only head fusion (args.fuse_heads / args.mfma_split): the flat parameter vector with
//   rgb_feature_linears.0.weight[:, :256] <- A W_rf,  .bias <- A b_rf + b_rh,   ins_feature_linears.0.weight <- W_ih W_if,  .bias <- W_ih b_if + b_ih
// Products accumulated in float64 and rounded ONCE (the fused weights are then as good as f32 weights can be); everything
// else is copied.  Once per weight update; replaces the torch float64 matmul (rocBLAS) the first version used.
__global__ void head_fuse_params_kernel(const float* __restrict__ flat, const Params P, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.total) return;
    float v = flat[e];
    for (int br = 0; br < 2; ++br) {
        const Lin& hid = br ? P.ins_hidden : P.rgb_hidden;
        const Lin& feat = br ? P.ins_feature : P.rgb_feature;
        if (e >= hid.w_off && e < hid.w_off + (int64_t)hid.out * hid.in) {
            const int i = (int)((e - hid.w_off) / hid.in), j = (int)((e - hid.w_off) % hid.in);
            if (j < W) {                                             // (the 27 direction columns of the rgb layer stay)
                const float* __restrict__ a = flat + hid.w_off + (int64_t)i * hid.in;
                double s = 0.0;
                for (int k = 0; k < W; ++k) s = __builtin_fma((double)a[k], (double)flat[feat.w_off + (int64_t)k * W + j], s);
                v = (float)s;
            }
        } else if (e >= hid.b_off && e < hid.b_off + hid.out) {
            const int i = (int)(e - hid.b_off);
            const float* __restrict__ a = flat + hid.w_off + (int64_t)i * hid.in;
            double s = (double)flat[e];
            for (int k = 0; k < W; ++k) s = __builtin_fma((double)a[k], (double)flat[feat.b_off + k], s);
            v = (float)s;
        }
    }
    out[e] = v;
}

// outputs, in this order: dA [128][256] | dW_rf [256][256] | db_rf [256] | dW_ih [128][256] | dW_if [256][256] | db_if [256]
constexpr int N_DA = HW * W, N_DW = W * W;
constexpr int UNFUSE_OUTPUTS = 2 * (N_DA + N_DW + W);

__global__ void head_unfuse_kernel(const float* __restrict__ flat, const Params P, const float* __restrict__ G,
                                   const float* __restrict__ Q, float* __restrict__ grad) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= UNFUSE_OUTPUTS) return;
    const bool ins = e >= N_DA + N_DW + W;
    if (ins) e -= N_DA + N_DW + W;
    const Lin& hid = ins ? P.ins_hidden : P.rgb_hidden;              // A (| B)  /  W_ih
    const Lin& feat = ins ? P.ins_feature : P.rgb_feature;           // W_rf     /  W_if
    const float* __restrict__ X = ins ? Q : G;                       // [128][256]
    const float* __restrict__ s = grad + hid.b_off;                  // s1 / s2: the hidden layer's bias gradient (already reduced)
    if (e < N_DA) {                                                  // dA[i][j] = sum_k X[i][k] Wf[j][k] + s[i] bf[j]
        const int i = e / W, j = e % W;
        const float* __restrict__ x = X + (int64_t)i * W;
        const float* __restrict__ wf = flat + feat.w_off + (int64_t)j * W;
        float acc = 0.f;
        for (int k = 0; k < W; ++k) acc = fmaf(x[k], wf[k], acc);
        acc = fmaf(s[i], flat[feat.b_off + j], acc);
        grad[hid.w_off + (int64_t)i * hid.in + j] = acc;
    } else if (e < N_DA + N_DW) {                                    // dWf[i][j] = sum_k Whid[k][i] X[k][j]
        e -= N_DA;
        const int i = e / W, j = e % W;
        const float* __restrict__ a = flat + hid.w_off + i;
        float acc = 0.f;
        for (int k = 0; k < HW; ++k) acc = fmaf(a[(int64_t)k * hid.in], X[(int64_t)k * W + j], acc);
        grad[feat.w_off + e] = acc;
    } else {                                                         // dbf[i] = sum_k Whid[k][i] s[k]
        const int i = e - N_DA - N_DW;
        const float* __restrict__ a = flat + hid.w_off + i;
        float acc = 0.f;
        for (int k = 0; k < HW; ++k) acc = fmaf(a[(int64_t)k * hid.in], s[k], acc);
        grad[feat.b_off + i] = acc;
    }
}

}  // namespace

extern "C" int dmnerf_head_product(const float* d_flat, int ins_num, float* d_F, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "head_product: ins_num %d unsupported", ins_num);
    if (!d_flat || !d_F) return dmn_fail(DMNERF_E_ARG, "head_product: null pointer");
    hipLaunchKernelGGL(head_product_kernel, dim3((HW * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_flat, make_params(ins_num), d_F);
    return dmn_check_launch("head_product");
}

extern "C" int dmnerf_fuse_heads(const float* d_flat, int ins_num, float* d_flat_fused, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "fuse_heads: ins_num %d unsupported", ins_num);
    if (!d_flat || !d_flat_fused || d_flat == d_flat_fused) return dmn_fail(DMNERF_E_ARG, "fuse_heads: null or aliased pointer");
    const Params P = make_params(ins_num);
    hipLaunchKernelGGL(head_fuse_params_kernel, dim3((unsigned)((P.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_flat, P, d_flat_fused);
    return dmn_check_launch("fuse_heads");
}

// (called by dmnerf_mlp_bwd_weights, wgrad.hip, after the reduction wrote G, Q and the two hidden bias gradients)
int dmn_head_unfuse(const float* d_flat, int ins_num, const float* d_G, const float* d_Q, float* d_grad, hipStream_t stream) {
    hipLaunchKernelGGL(head_unfuse_kernel, dim3((UNFUSE_OUTPUTS + 255) / 256), dim3(256), 0, stream, d_flat, make_params(ins_num), d_G, d_Q, d_grad);
    return dmn_check_launch("mlp_bwd_weights: head_unfuse");
}
