// mlp_bwd.hip -- data-gradient (dgrad) pass of the DM-NeRF MLP for gfx950.
//
// Autograd of DM_NeRF.forward (networks/dm_nerf.py:80-106) w.r.t. every pre-activation, given
// dL/draw [M, 4+C]: the same register-chained structure as the forward (mlp_fwd.hip) with W^T
// as the MFMA A operand -- the accumulator layout of dy_l is the B operand of W_l^T.  One wave
// owns 32 samples; ReLU masks come from the activations the forward saved; every dy is written
// feature-major [rows][M] next to the saved inputs, which is exactly what the weight-gradient
// GEMMs (dW = dy . x^T over the M samples) consume.
//
// Gradient barriers of the reference are honoured structurally:
//   * ins branch input is h.detach() (dm_nerf.py:95): dq is NOT added to dh_7;
//   * no gradient flows to the encodings (rays / depths are not parameters).
// 9280 MFMAs per 32 samples at C=14 (dgrad MACs = fwd MACs - 101 248).
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "layout.h"
#include "mlp_common.h"

using namespace dmn;

namespace {

struct BwdArgs {
    const float* blob;     // forward blob (density / rgb_linear VALU segments)
    const float* blobT;    // transposed segments (layout.h::BlobTLayout)
    BlobLayout L;
    BlobTLayout LT;
    const float* save;     // forward activations (SaveLayout)
    const float* graw;     // [M, 4+C]
    float* dsave;          // gradients, same SaveLayout (rows of pe/de unused)
    int64_t M;
};

template <int NB>
__device__ __forceinline__ void mask_relu(f32x16 (&d)[NB], const f32x16 (&act)[NB], const f32x16 (&g)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[b][r] = act[b][r] > 0.f ? g[b][r] : 0.f;      // relu backward: grad * (result > 0)
}

template <int NB>
__device__ __forceinline__ void zero(f32x16 (&v)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = (f32x16)(0.f);
}

template <int OBI>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(const BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk * 32 >= a.M) return;
    const int64_t m_raw = blk * 32 + (lane & 31);
    const bool valid = m_raw < a.M;
    const int64_t m = valid ? m_raw : a.M - 1;

    const BlobLayout& L = a.L;
    const BlobTLayout& LT = a.LT;
    const rsrc_t rsF = make_rsrc(a.blob, L.total);
    const rsrc_t rsT = make_rsrc(a.blobT, LT.total);
    const int voff = lane * 16;
    const SaveLayout SL = make_save_layout(a.M);
    const int64_t M = save_row_len(a.M);            // padded row length (see layout.h)

    // ---- incoming gradient of this lane's sample --------------------------------------------
    const float* __restrict__ gr = a.graw + m * (4 + L.C);
    // tail lanes get a zero gradient: every dy they produce is then exactly 0, so the padding
    // columns contribute nothing to the weight / bias gradients
    const float g_rgb[3] = {valid ? gr[0] : 0.f, valid ? gr[1] : 0.f, valid ? gr[2] : 0.f};
    const float g_sigma = valid ? gr[3] : 0.f;
    f32x16 gi[OBI];
#pragma unroll
    for (int b = 0; b < OBI; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
            gi[b][r] = (ch < L.C && valid) ? gr[4 + ch] : 0.f;
        }

    f32x16 d[8], acc[8];
    {
        // ---- ins branch: dg2 = relu'(g2) . (W_io^T g_ins);  dq = W_ih^T dg2 ----------------------
        f32x16 t4[4], act4[4], d4[4];
        zero<4>(t4);
        gemm_seg<4 * OBI, 4, OBI>(rsT, (int)LT.t_inso, gi, t4, voff);
        load_rows<4>(make_rowio(a.save + SL.g2, 128, M, blk, lane), act4);
        mask_relu<4>(d4, act4, t4);
        store_rows<4>(make_rowio(a.dsave + SL.g2, 128, M, blk, lane), d4);
        zero<8>(acc);
        gemm_seg<16, 8, 4>(rsT, (int)LT.t_insh, d4, acc, voff);
        store_rows<8>(make_rowio(a.dsave + SL.q, 256, M, blk, lane), acc);      // dq (ins_feature has no activation)

        // ---- rgb branch: dg1 = relu'(g1) . (W_ro^T g_rgb) on the VALU;  df = (W_rh^T dg1)[:256] ----
        zero<4>(t4);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 w = ldw(rsF, half * 256, ((int)L.w_rgbo + c * 128 + 4 * i) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = 4 * i + j;
                    t4[p >> 4][p & 15] = fmaf(w[j], g_rgb[c], t4[p >> 4][p & 15]);
                }
            }
        }
        load_rows<4>(make_rowio(a.save + SL.g1, 128, M, blk, lane), act4);
        mask_relu<4>(d4, act4, t4);
        store_rows<4>(make_rowio(a.dsave + SL.g1, 128, M, blk, lane), d4);
        zero<8>(acc);
        gemm_seg<16, 8, 4>(rsT, (int)LT.t_rgbh, d4, acc, voff);
        store_rows<8>(make_rowio(a.dsave + SL.f, 256, M, blk, lane), acc);      // df (rgb_feature has no activation)
#pragma unroll
        for (int b = 0; b < 8; ++b) d[b] = acc[b];
    }

    // ---- trunk: st = 0: dh_7 = W_rf^T df + w_d g_sigma;  st = k: dh_{7-k} = W_{8-k}^T dy_{8-k} -----
#pragma nounroll
    for (int st = 0; st < NSTAGE_T; ++st) {
        zero<8>(acc);
        gemm_seg<32, 8, 8>(rsT, (int)LT.t_stage + st * (int)seg_floats(32, 8), d, acc, voff);
        if (st == 0) {
            // density_linear (dm_nerf.py:101): dh_7 += w_d * g_sigma
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const f32x4 w = ldw(rsF, half * 512, ((int)L.w_den + 4 * i) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = 4 * i + j;
                    acc[p >> 4][p & 15] = fmaf(w[j], g_sigma, acc[p >> 4][p & 15]);
                }
            }
        }
        const int l = 7 - st;                                   // layer whose pre-activation gradient this is
        f32x16 act[8];
        load_rows<8>(make_rowio(a.save + SL.h + (int64_t)l * 256 * M, 256, M, blk, lane), act);
        mask_relu<8>(d, act, acc);
        store_rows<8>(make_rowio(a.dsave + SL.h + (int64_t)l * 256 * M, 256, M, blk, lane), d);
    }
}

}  // namespace

extern "C" int dmnerf_mlp_bwd_data(const float* d_blob, const float* d_blob_t, int ins_num, const float* d_save,
                                   const float* d_graw, int64_t M, float* d_dsave, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data: ins_num %d unsupported", ins_num);
    if (M < 0 || M > DMNERF_MAX_TRAIN_SAMPLES) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data: M=%lld outside [0,%lld]", (long long)M, (long long)DMNERF_MAX_TRAIN_SAMPLES);
    if (M == 0) return DMNERF_OK;
    if (!d_blob || !d_blob_t || !d_save || !d_graw || !d_dsave) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data: null pointer");
    BwdArgs a{};
    a.blob = d_blob; a.blobT = d_blob_t; a.L = make_layout(ins_num); a.LT = make_layout_t(ins_num);
    a.save = d_save; a.graw = d_graw; a.dsave = d_dsave; a.M = M;
    const int64_t nblk = (M + 31) / 32;
    dim3 g((unsigned)((nblk + 3) / 4)), b(256);
    switch (a.L.OBI) {
        case 1: hipLaunchKernelGGL(mlp_bwd_kernel<1>, g, b, 0, (hipStream_t)stream, a); break;
        case 2: hipLaunchKernelGGL(mlp_bwd_kernel<2>, g, b, 0, (hipStream_t)stream, a); break;
        case 3: hipLaunchKernelGGL(mlp_bwd_kernel<3>, g, b, 0, (hipStream_t)stream, a); break;
        case 4: hipLaunchKernelGGL(mlp_bwd_kernel<4>, g, b, 0, (hipStream_t)stream, a); break;
        default: return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data: unsupported logit count C=%d", a.L.C);
    }
    return dmn_check_launch("mlp_bwd_data");
}
