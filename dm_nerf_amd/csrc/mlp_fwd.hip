// mlp_fwd.hip -- fused "points -> positional encoding -> DM-NeRF MLP" forward for gfx950.
//
// Replaces networks/render.py:49-61 / :71-83 (pts, embed x2, cat) + DM_NeRF.forward
// (networks/dm_nerf.py:80-106): 11 nn.Linear + ReLU + 3 cats per sample.
//
// Design: one wave owns 32 samples for the whole network.  Activations
// never leave registers: with Y^T = W . X^T on v_mfma_f32_32x32x2_f32 (A = weights, B = X^T),
// the accumulator layout of layer n IS the B-operand layout of layer n+1 (see layout.h), so a
// layer is 8 x 128 back-to-back MFMAs whose only memory traffic is the pre-permuted weight
// stream (one coalesced dwordx4 per lane per 4 MFMAs, L2-resident: a model is 2.8 MB).
// Exact f32: the MFMA is bitwise an fmaf chain in k order (MI355X guide), so results are in the
// f32-roundoff class of the reference's sgemm.
//
// Roofline: MFMA f32 (157.3 TFLOP/s).  10880 MFMAs per 32 samples at C=14 vs 10836.0 ideal (99.6 %).
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "layout.h"
#include "mlp_common.h"

using namespace dmn;

namespace {

struct MlpArgs {
    const float* blob;
    BlobLayout L;
    const float* rays_o;   // rays variant
    const float* rays_d;
    const float* z;
    const float* x;        // embedded variant [M, 90]
    float* raw;            // [M, 4+C]
    float* save;           // training: activation workspace, SAVE_ROWS x M floats (layout.h::SaveLayout)
    int64_t M;             // total samples
    int S;                 // samples per ray (rays variant)
};

template <int OBI, bool EMBEDDED, bool SAVE>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(const MlpArgs a) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);     // 32-sample block of this wave
    if (blk * 32 >= a.M) return;                                          // wave-uniform
    const int64_t m_raw = blk * 32 + (lane & 31);
    const bool valid = m_raw < a.M;
    const int64_t m = valid ? m_raw : a.M - 1;                            // tail lanes recompute the last sample

    const float* __restrict__ blob = a.blob;
    const BlobLayout& L = a.L;
    const rsrc_t rs = make_rsrc(blob, L.total);
    const int voff = lane * 16;     // per-lane byte offset inside a 1 KiB (64 x float4) weight row
    const int hoff = half * 64;     // per-half byte offset inside a bias row pair

    f32x16 pe[2];   // 32 k-pairs of the position encoding (63 columns + pad)
    f32x16 de[1];   // 16 k-pairs of the direction encoding (27 columns + pad)
    if constexpr (EMBEDDED) {
        const float* xr = a.x + m * (POS_CH + DIR_CH);
        load_encoded<POS_L, 2>(xr, pe, half);
        load_encoded<DIR_L, 1>(xr + POS_CH, de, half);
    } else {
        const int64_t n = m / a.S;
        const float ox = a.rays_o[n * 3 + 0], oy = a.rays_o[n * 3 + 1], oz = a.rays_o[n * 3 + 2];
        const float dx = a.rays_d[n * 3 + 0], dy = a.rays_d[n * 3 + 1], dz = a.rays_d[n * 3 + 2];
        const float zv = a.z[m];
        // pts = rays_o + rays_d * z   (render.py:49: separate multiply and add, no fma)
        const float pt[3] = {ox + dx * zv, oy + dy * zv, oz + dz * zv};
        // viewdirs = rays_d / ||rays_d||   (render.py:37)
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        const float vd[3] = {dx / nrm, dy / nrm, dz / nrm};
        encode<POS_L, 2>(pt, pe, half);
        encode<DIR_L, 1>(vd, de, half);
    }

    const SaveLayout SL = make_save_layout(a.M);
    const int64_t MP = save_row_len(a.M);          // padded row length of the training workspace
    if constexpr (SAVE) {                          // tail lanes write the padding columns of the last block
        store_encoded_rows<POS_L, 2>(a.save + SL.pe, MP, blk, lane, pe);
        store_encoded_rows<DIR_L, 1>(a.save + SL.de, MP, blk, lane, de);
    }

    f32x16 h[8], acc[8];
    // mlps.0 : 63 -> 256
    init_bias<8>(rs, (int)L.b0, acc, hoff);
    gemm_seg<8, 8, 2>(rs, (int)L.w0, pe, acc, voff);
#pragma unroll
    for (int b = 0; b < 8; ++b) h[b] = relu16(acc[b]);
    // Activation saves are issued as one burst per layer.  (Interleaving them into the next layer's
    // MFMA stream was measured SLOWER, 14.3 vs 12.2 ms: vmcnt retires in order, so every later
    // weight-load wait then also waits for a store acknowledgement.)
    if constexpr (SAVE) store_rows<8>(make_rowio(a.save + SL.h, 256, MP, blk, lane), h);

    float sigma = 0.f, rgb_out[3] = {0.f, 0.f, 0.f};
    float* __restrict__ out_row = a.raw + m * (4 + L.C);

#pragma nounroll
    for (int st = 0; st < NSTAGE; ++st) {
        init_bias<8>(rs, (int)L.b_stage + st * (int)bias_floats(8), acc, hoff);
        gemm_seg<32, 8, 8>(rs, (int)L.w_stage + st * (int)seg_floats(32, 8), h, acc, voff);
        if (st == 4) gemm_seg<8, 8, 2>(rs, (int)L.w5pe, pe, acc, voff);   // skip: cat[h, pts] (dm_nerf.py:87)
        if (st < 7) {
#pragma unroll
            for (int b = 0; b < 8; ++b) h[b] = relu16(acc[b]);
            if constexpr (SAVE) store_rows<8>(make_rowio(a.save + SL.h + (int64_t)(st + 1) * 256 * MP, 256, MP, blk, lane), h);
            if (st == 6) {
                // density_linear(h) (dm_nerf.py:101) on the VALU: 128 features per lane + the other half
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const f32x4 w = ldw(rs, half * 512, ((int)L.w_den + 4 * i) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int p = 4 * i + j;
                        part = fmaf(h[p >> 4][p & 15], w[j], part);
                    }
                }
                sigma = part + __shfl_xor(part, 32) + blob[L.b_den];
            }
        } else if (st == 7) {
            // acc = rgb_feature (no activation, dm_nerf.py:89); hidden = relu(W [rgb_feature, dirs]) (:90-93)
            f32x16 hid[4];
            init_bias<4>(rs, (int)L.b_rgbh, hid, hoff);
            if constexpr (SAVE) store_rows<8>(make_rowio(a.save + SL.f, 256, MP, blk, lane), acc);
            gemm_seg<32, 4, 8>(rs, (int)L.w_rgbh, acc, hid, voff);
            gemm_seg<4, 4, 1>(rs, (int)L.w_rgbh_dir, de, hid, voff);
#pragma unroll
            for (int b = 0; b < 4; ++b) hid[b] = relu16(hid[b]);
            if constexpr (SAVE) store_rows<4>(make_rowio(a.save + SL.g1, 128, MP, blk, lane), hid);
            // rgb_linear (dm_nerf.py:102) on the VALU
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x4 w = ldw(rs, half * 256, ((int)L.w_rgbo + c * 128 + 4 * i) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int p = 4 * i + j;
                        part = fmaf(hid[p >> 4][p & 15], w[j], part);
                    }
                }
                rgb_out[c] = part + __shfl_xor(part, 32) + blob[L.b_rgbo + c];
            }
        } else {
            // acc = ins_feature (input h.detach(), dm_nerf.py:95-96); hidden = relu(W ins_feature) (:97-99)
            f32x16 hid[4];
            init_bias<4>(rs, (int)L.b_insh, hid, hoff);
            if constexpr (SAVE) store_rows<8>(make_rowio(a.save + SL.q, 256, MP, blk, lane), acc);
            gemm_seg<32, 4, 8>(rs, (int)L.w_insh, acc, hid, voff);
#pragma unroll
            for (int b = 0; b < 4; ++b) hid[b] = relu16(hid[b]);
            if constexpr (SAVE) store_rows<4>(make_rowio(a.save + SL.g2, 128, MP, blk, lane), hid);
            f32x16 io[OBI];
            init_bias<OBI>(rs, (int)L.b_inso, io, hoff);
            gemm_seg<16, OBI, 4>(rs, (int)L.w_inso, hid, io, voff);     // ins_linear (:103)
            if (valid) {
#pragma unroll
                for (int b = 0; b < OBI; ++b) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (ch < L.C) out_row[4 + ch] = io[b][r];
                    }
                }
            }
        }
    }
    // cat[rgb, density, ins]  (dm_nerf.py:105)
    if (valid && half == 0) {
        out_row[0] = rgb_out[0];
        out_row[1] = rgb_out[1];
        out_row[2] = rgb_out[2];
        out_row[3] = sigma;
    }
}

template <bool EMBEDDED, bool SAVE>
int launch(const MlpArgs& a, hipStream_t stream) {
    const int64_t nblk = (a.M + 31) / 32;
    const int64_t grid = (nblk + 3) / 4;
    if (grid > 0x7fffffffLL) return dmn_fail(DMNERF_E_ARG, "mlp_fwd: %lld samples is too many for one launch", (long long)a.M);
    if (SAVE && a.M > DMNERF_MAX_TRAIN_SAMPLES)
        return dmn_fail(DMNERF_E_ARG, "mlp_fwd_train: %lld samples per launch exceed %lld (32-bit row offsets); split the batch",
                        (long long)a.M, (long long)DMNERF_MAX_TRAIN_SAMPLES);
    dim3 g((unsigned)grid), b(256);
    switch (a.L.OBI) {
        case 1: hipLaunchKernelGGL((mlp_fwd_kernel<1, EMBEDDED, SAVE>), g, b, 0, stream, a); break;
        case 2: hipLaunchKernelGGL((mlp_fwd_kernel<2, EMBEDDED, SAVE>), g, b, 0, stream, a); break;
        case 3: hipLaunchKernelGGL((mlp_fwd_kernel<3, EMBEDDED, SAVE>), g, b, 0, stream, a); break;
        case 4: hipLaunchKernelGGL((mlp_fwd_kernel<4, EMBEDDED, SAVE>), g, b, 0, stream, a); break;
        default: return dmn_fail(DMNERF_E_ARG, "mlp_fwd: unsupported logit count C=%d", a.L.C);
    }
    return dmn_check_launch("mlp_fwd");
}

}  // namespace

extern "C" int dmnerf_mlp_fwd_embedded(const float* d_blob, int ins_num, const float* d_x, int64_t M,
                                       float* d_raw, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_embedded: ins_num %d unsupported", ins_num);
    if (M < 0) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_embedded: M < 0");
    if (M == 0) return DMNERF_OK;      // an empty batch is legal (and has null data pointers)
    if (!d_blob || !d_x || !d_raw) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_embedded: null pointer");
    MlpArgs a{};
    a.blob = d_blob; a.L = make_layout(ins_num); a.x = d_x; a.raw = d_raw; a.M = M; a.S = 1;
    return launch<true, false>(a, (hipStream_t)stream);
}

extern "C" int dmnerf_mlp_fwd_rays(const float* d_blob, int ins_num, const float* d_rays_o,
                                   const float* d_rays_d, const float* d_z, int64_t N, int S,
                                   float* d_raw, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays: ins_num %d unsupported", ins_num);
    if (N < 0 || S < 1) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays: bad N=%lld S=%d", (long long)N, S);
    if (N == 0) return DMNERF_OK;
    if (!d_blob || !d_rays_o || !d_rays_d || !d_z || !d_raw) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays: null pointer");
    MlpArgs a{};
    a.blob = d_blob; a.L = make_layout(ins_num); a.rays_o = d_rays_o; a.rays_d = d_rays_d; a.z = d_z;
    a.raw = d_raw; a.M = N * S; a.S = S;
    return launch<false, false>(a, (hipStream_t)stream);
}

extern "C" int64_t dmnerf_train_save_floats(int64_t M) { return M < 0 ? -1 : make_save_layout(M).total; }

extern "C" int dmnerf_mlp_fwd_rays_train(const float* d_blob, int ins_num, const float* d_rays_o,
                                         const float* d_rays_d, const float* d_z, int64_t N, int S,
                                         float* d_raw, float* d_save, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train: ins_num %d unsupported", ins_num);
    if (N < 0 || S < 1) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train: bad N=%lld S=%d", (long long)N, S);
    if (N == 0) return DMNERF_OK;
    if (!d_blob || !d_rays_o || !d_rays_d || !d_z || !d_raw || !d_save) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train: null pointer");
    MlpArgs a{};
    a.blob = d_blob; a.L = make_layout(ins_num); a.rays_o = d_rays_o; a.rays_d = d_rays_d; a.z = d_z;
    a.raw = d_raw; a.save = d_save; a.M = N * S; a.S = S;
    return launch<false, true>(a, (hipStream_t)stream);
}
