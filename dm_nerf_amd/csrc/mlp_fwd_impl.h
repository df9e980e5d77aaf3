// mlp_fwd_impl.h -- fused "points -> positional encoding -> DM-NeRF MLP" forward kernel for gfx950 (entry points: mlp_fwd*.hip).
//
// Replaces networks/render.py:49-61 / :71-83 (pts, embed x2, cat) + DM_NeRF.forward
// (networks/dm_nerf.py:80-106): 11 nn.Linear + ReLU + 3 cats per sample.
//
// Design: one wave owns 32 samples for the whole network.  Activations
// never leave registers: with Y^T = W . X^T on v_mfma_f32_32x32x2_f32 (A = weights, B = X^T),
// the accumulator layout of layer n IS the B-operand layout of layer n+1 (see layout.h), so a
// layer is 8 x 128 back-to-back MFMAs whose only memory traffic is the pre-permuted weight
// stream, which the workgroup's 4 waves pull through a 2 x 64 KiB LDS ring by LDS-DMA two
// quarters ahead of use (mlp_common.h): A operands are ds_read_b128, no VMEM load in the MFMA stream.
// Exact f32: the MFMA is bitwise an fmaf chain in k order (MI355X guide), so results are in the
// f32-roundoff class of the reference's sgemm.
//
// Roofline: MFMA f32 (157.3 TFLOP/s).  10880 MFMAs per 32 samples at C=14 vs 10836.0 ideal (99.6 %).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "layout.h"
#include "mlp_common.h"

#ifdef DMN_FWD_TRACE
extern long long* g_dmn_fwd_trace;        // diagnostic builds (make diag); defined in mlp_fwd.hip
#endif

using namespace dmn;

namespace {

constexpr int PARK_FLOATS = 4096;     // 16 KiB behind ring + table: 4 KiB per wave, the parked direction encoding

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
#ifdef DMN_FWD_TRACE
    long long* trace;      // diagnostic builds only (make diag): per-workgroup cycle stamps, see scripts/diag_fwd.py
#endif
};
#ifdef DMN_FWD_TRACE
#define DMN_STAMP(k) do { if (a.trace && threadIdx.x == 0) { a.trace[8 * blockIdx.x + (k)] = (long long)clock64(); if ((k) == 0) a.trace[8 * blockIdx.x + 7] = (long long)wall_clock64(); } } while (0)
#else
#define DMN_STAMP(k) do {} while (0)
#endif

template <int OBI, bool EMBEDDED, bool SAVE, bool FUSED = false>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(const MlpArgs a) {
    // (FUSED && SAVE = the opt-in training forward on the fused-heads blob: the backward is in re-associated form anyway,
    // csrc/heads.hip, and needs exactly what this variant saves: pe, de, h_0..h_7, g1, g2 and the masks)
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [ring 2 x 64 KiB][table 16 KiB][park 16 KiB]
    float* const tab = lds + RING_FLOATS;
    DMN_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Every wave of the workgroup takes part in the DMA + barrier protocol; a wave beyond the end of
    // the batch (only in the last workgroup) is an exact duplicate of the last block's wave: it computes and
    // stores the same values to the same addresses.
    const int64_t nblk = (a.M + 31) / 32;
    const int64_t blk_raw = (int64_t)blockIdx.x * 4 + wave;
    const bool wave_active = blk_raw < nblk;
    const int64_t blk = wave_active ? blk_raw : nblk - 1;
    // (recomputed where the outputs are written: blk is wave-uniform and the lane id is free, so nothing of this has
    // to stay live in VGPRs across the network)
    // `fresh` hides a value's origin from the optimizer: an address or index derived from fresh(lane) / fresh(half) is
    // computed where it is used instead of being hoisted to the top of the kernel and carried (spilled) across the network
    auto fresh = [](int x) -> int { asm volatile("" : "+v"(x)); return x; };
    auto sample_of_lane = [&]() -> int64_t { return blk * 32 + (fresh(lane) & 31); };
    const int64_t m_in = blk * 32 + (lane & 31);
    const int64_t m = m_in < a.M ? m_in : a.M - 1;                           // tail lanes recompute the last sample
    // The direction encoding (16 registers) is needed once, ~4400 MFMAs from here, by the rgb hidden layer: it is parked in
    // this wave's 4 KiB of the last 16 KiB of the CU's LDS instead of occupying VGPRs through the whole trunk (with it
    // resident the training variants spilled ~30 registers to scratch around the heads).
    auto park = [&]() -> f32x4* { return reinterpret_cast<f32x4*>(lds + LDS_FLOATS + wave * (PARK_FLOATS / 4)) + lane; };

    const float* __restrict__ blob = a.blob;
    const BlobLayout& L = a.L;

    // ---- inputs first (their loads are the oldest VMEM ops), then the table, then the first two quarters
    float pt[3], vd[3];
    const float* xr = nullptr;
    if constexpr (EMBEDDED) {
        xr = a.x + m * (POS_CH + DIR_CH);
    } else {
        const int64_t n = m / a.S;
        const float ox = a.rays_o[n * 3 + 0], oy = a.rays_o[n * 3 + 1], oz = a.rays_o[n * 3 + 2];
        const float dx = a.rays_d[n * 3 + 0], dy = a.rays_d[n * 3 + 1], dz = a.rays_d[n * 3 + 2];
        const float zv = a.z[m];
        // pts = rays_o + rays_d * z   (render.py:49: separate multiply and add, no fma)
        pt[0] = ox + dx * zv; pt[1] = oy + dy * zv; pt[2] = oz + dz * zv;
        // viewdirs = rays_d / ||rays_d||   (render.py:37)
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        vd[0] = dx / nrm; vd[1] = dy / nrm; vd[2] = dz / nrm;
    }
    f32x16 pe[2];   // 32 k-pairs of the position encoding (63 columns + pad)
    f32x16 de[1];   // 16 k-pairs of the direction encoding (27 columns + pad)
    if constexpr (EMBEDDED) {
        load_encoded<POS_L, 2>(xr, pe, half);
        load_encoded<DIR_L, 1>(xr + POS_CH, de, half);
    }
    // biases + VALU heads: global -> registers now, registers -> LDS table after the encoding below, so
    // that neither this load nor the first weight DMA exposes its latency
    f32x4 tabv[TAB_FLOATS / 1024];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(blob) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < TAB_FLOATS / 1024; ++k) tabv[k] = src[k * 256];
    }
    WStream ws;
    ws_init(ws, blob, L.total, lds, lane, wave, L.stream);
    ws_fetch_first(ws);                                                   // quarter 0: mlps.0
    if constexpr (!EMBEDDED) {                                            // full-range sin/cos under the DMA flight
        encode<POS_L, 2>(pt, pe, half);
        encode<DIR_L, 1>(vd, de, half);
    }
    {
        f32x4* dst = reinterpret_cast<f32x4*>(tab) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < TAB_FLOATS / 1024; ++k) dst[k * 256] = tabv[k];
    }

    const SaveLayout SL = make_save_layout(a.M);
    const int64_t MP = save_row_len(a.M);          // padded row length of the training workspace
    const int srows = 1;                           // (every wave stores: see mlp_common.h::RowIO)
    // ReLU bit masks for the backward pass (one 16-byte store per lane and layer instead of 128 row loads there)
    rsrc_t bits_rs;
    int bits_voff = 0;
    if constexpr (SAVE) {
        bits_rs = uniform_rsrc(a.save + SL.bits, (int64_t)srows * (BITS_WORDS_PER_BLOCK / 32) * MP);
        bits_voff = (int)((blk * BITS_WORDS_PER_BLOCK + lane * 4) * 4);
    }

    DMN_STAMP(1);
    f32x16 h[8], acc[8];
    auto save_mask8 = [&](int layer) {
        unsigned m[4];
        pack_mask<8>(h, m);
        u32x4 v = {m[0], m[1], m[2], m[3]};
        __builtin_amdgcn_raw_buffer_store_b128(v, bits_rs, bits_voff + layer * 1024, 0, 0);
    };
    // ---- mlps.0 : 63 -> 256 (quarter 0)
    ws_prime<8>(ws, lane);
    if constexpr (SAVE) {                          // 90 stores with a whole quarter to retire
        store_encoded_rows<POS_L, 2>(a.save + SL.pe, srows * MP, blk, lane, pe);
        store_encoded_rows<DIR_L, 1>(a.save + SL.de, srows * MP, blk, lane, de);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {                  // park the direction encoding (conflict-free: 16 B per lane, lane-major)
        const f32x4 v = {de[0][4 * q + 0], de[0][4 * q + 1], de[0][4 * q + 2], de[0][4 * q + 3]};
        park()[q * 64] = v;
    }
    init_bias_lds<8>(tab + L.b0, acc, half);
    gemm_quarter<0, 8, 8, 8>(ws, pe, acc, lane);
#pragma unroll
    for (int b = 0; b < 8; ++b) h[b] = relu16(acc[b]);
    if constexpr (SAVE) save_mask8(0);

    DMN_STAMP(2);
    float sigma = 0.f, rgb_out[3] = {0.f, 0.f, 0.f};

    // One 256 -> 256 stage = 4 quarters.  Stage st consumes h_st (st = 0..7; st = 8 re-reads h_7): training
    // saves it with stores spread over the MFMA gaps of quarters 1..3, younger than each quarter's DMA pieces
    // (see mlp_common.h).  The body is instantiated three times (trunk loop, rgb_feature, ins_feature) to keep
    // the register live ranges of the two heads out of the loop.
    auto stage = [&](int st, auto next_ob, auto save_h) {   // next_ob: out-blocks of the quarter after the stage; save_h: store its input
        constexpr int NEXT = decltype(next_ob)::value;
        constexpr bool SAVE_H = SAVE && decltype(save_h)::value;
        RowIO hio;
        if constexpr (SAVE) hio = make_rowio(a.save + SL.h + (int64_t)(st < 8 ? st : 7) * 256 * MP, 256, (st < 8 ? srows : 0) * MP, blk, lane);
        init_bias_lds<8>(tab + L.b_stage + st * (int)bias_floats(8), acc, half);
        // training: the stage's input h is saved while it is consumed, 43 + 43 + 42 TID-addressed stores riding in
        // the MFMA gaps of quarters 1..3 (a burst of stores stalls the one wave; spread out they cost nothing)
        constexpr int NS = SAVE_H ? 43 : 0;
        auto st_h = [&](int k0) { return [&, k0](int k) { store_row_one(hio, h, k0 + k); }; };
        gemm_quarter<0, 8, 8, 8>(ws, h, acc, lane);
        gemm_quarter<8, 8, 8, 8, false, NS>(ws, h, acc, lane, st_h(0));
        gemm_quarter<16, 8, 8, 8, false, NS>(ws, h, acc, lane, st_h(43));
        gemm_quarter<24, 8, 8, NEXT, false, SAVE_H ? 42 : 0>(ws, h, acc, lane, st_h(86));
    };
    typedef std::integral_constant<int, 8> Next8;
    typedef std::integral_constant<int, 4> Next4;

    // ---- trunk: mlps.1 .. mlps.7
#pragma nounroll
    for (int st = 0; st < 7; ++st) {
        stage(st, Next8{}, std::true_type{});
        if (st == 4) {                                                    // skip: cat[h, pts] (dm_nerf.py:87)
            gemm_quarter<0, 8, 8, 8>(ws, pe, acc, lane);
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) h[b] = relu16(acc[b]);
        if constexpr (SAVE) save_mask8(st + 1);
    }
    {
        // density_linear(h) (dm_nerf.py:101) on the VALU: 128 features per lane + the other half
        const f32x4* wd = reinterpret_cast<const f32x4*>(tab + L.w_den + fresh(half) * 128);
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const f32x4 w = wd[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = 4 * i + j;
                part = fmaf(h[p >> 4][p & 15], w[j], part);
            }
        }
        sigma = part + __shfl_xor(part, 32) + tab[L.b_den];
        asm volatile("" : "+v"(sigma));      // finished HERE: otherwise the tail of the dot product is sunk below the rgb branch
    }                                        // and its 24 weight registers are spilled across it

    DMN_STAMP(3);
    // ---- rgb branch: acc = rgb_feature (no activation, dm_nerf.py:89); hidden = relu(W [rgb_feature, dirs]) (:90-93)
    // FUSED: rgb_feature_linear is folded into rgb_feature_linears.0 (blob built by the caller): h feeds the hidden layer directly
    if constexpr (!FUSED) stage(7, Next4{}, std::true_type{});
    {
        f32x16 hid[4];
        init_bias_lds<4>(tab + L.b_rgbh, hid, half);
        // (training does not save rgb_feature / ins_feature: the backward folds the activation-free feature linears,
        // layout.h::BlobTLayout -- their weight gradients come from G = dg1 . h_7^T and Q = dg2 . h_7^T)
        auto& fsrc = *(FUSED ? &h : &acc);                             // the hidden layer's input: rgb_feature, or h itself when fused
        // fused training: there is no rgb_feature stage to save h_7 under -- it rides here, while it is this GEMM's B operand
        constexpr bool SAVE_H7 = SAVE && FUSED;
        RowIO h7io;
        if constexpr (SAVE_H7) h7io = make_rowio(a.save + SL.h + (int64_t)7 * 256 * MP, 256, srows * MP, blk, lane);
        auto st_h7 = [&](int k0) { return [&, k0](int k) { store_row_one(h7io, h, k0 + k); }; };
        gemm_quarter<0, 16, 4, 4, false, SAVE_H7 ? 63 : 0>(ws, fsrc, hid, lane, st_h7(0));
        gemm_quarter<16, 16, 4, 4, false, SAVE_H7 ? 63 : 0>(ws, fsrc, hid, lane, st_h7(63));
        f32x16 dpk[1];                                                 // the parked direction encoding comes back for its one quarter
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = park()[q * 64];
            dpk[0][4 * q + 0] = v[0]; dpk[0][4 * q + 1] = v[1]; dpk[0][4 * q + 2] = v[2]; dpk[0][4 * q + 3] = v[3];
        }
        gemm_quarter<0, 4, 4, 8, false, SAVE_H7 ? 2 : 0>(ws, dpk, hid, lane, st_h7(126));
#pragma unroll
        for (int b = 0; b < 4; ++b) hid[b] = relu16(hid[b]);
        if constexpr (SAVE) {
            store_rows<4>(make_rowio(a.save + SL.g1, 128, srows * MP, blk, lane), hid);
            unsigned m[2];
            pack_mask<4>(hid, m);
            __builtin_amdgcn_raw_buffer_store_b32(m[0], bits_rs, (int)((blk * BITS_WORDS_PER_BLOCK + 2048 + lane * 2) * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(m[1], bits_rs, (int)((blk * BITS_WORDS_PER_BLOCK + 2048 + lane * 2 + 1) * 4), 0, 0);
        }
        // rgb_linear (dm_nerf.py:102) on the VALU
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4* wr = reinterpret_cast<const f32x4*>(tab + L.w_rgbo + (c * 2 + fresh(half)) * 64);
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 w = wr[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = 4 * i + j;
                    part = fmaf(hid[p >> 4][p & 15], w[j], part);
                }
            }
            rgb_out[c] = part + __shfl_xor(part, 32) + tab[L.b_rgbo + c];
            asm volatile("" : "+v"(rgb_out[c]));
        }
    }

    // ---- ins branch: acc = ins_feature (input h.detach(), dm_nerf.py:95-96); hidden = relu(W ins_feature) (:97-99)
    if constexpr (!FUSED) stage(8, Next4{}, std::false_type{});          // re-reads h_7: already saved
    {
        f32x16 hid[4];
        init_bias_lds<4>(tab + L.b_insh, hid, half);
        auto& qsrc = *(FUSED ? &h : &acc);
        gemm_quarter<0, 16, 4, 4>(ws, qsrc, hid, lane);
        gemm_quarter<16, 16, 4, OBI>(ws, qsrc, hid, lane);
#pragma unroll
        for (int b = 0; b < 4; ++b) hid[b] = relu16(hid[b]);
        // training: the hidden layer (g2) is saved under the ins_linear quarter (43 / 63 spread stores), the rest as one short burst
        constexpr int NS3 = SAVE ? (OBI == 1 ? 43 : 63) : 0;          // side slots of the ins_linear quarter
        RowIO g2io;
        if constexpr (SAVE) {
            g2io = make_rowio(a.save + SL.g2, 128, srows * MP, blk, lane);
            store_rows_part<NS3, 64 - NS3>(g2io, hid);
            unsigned m[2];
            pack_mask<4>(hid, m);
            __builtin_amdgcn_raw_buffer_store_b32(m[0], bits_rs, (int)((blk * BITS_WORDS_PER_BLOCK + 2176 + lane * 2) * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(m[1], bits_rs, (int)((blk * BITS_WORDS_PER_BLOCK + 2176 + lane * 2 + 1) * 4), 0, 0);
        }
        f32x16 io[OBI];
        init_bias_lds<OBI>(tab + L.b_inso, io, half);
        auto st_3 = [&](int k) { store_row_one(g2io, hid, k); };
        gemm_quarter<0, 16, OBI, 0, false, NS3>(ws, hid, io, lane, st_3);   // ins_linear (:103); its fetch runs into the zero-filled landing zone
        const int64_t ms = sample_of_lane();
        const int hf = fresh(half);
        float* __restrict__ out_row = a.raw + (ms < a.M ? ms : a.M - 1) * (4 + L.C);
        if (ms < a.M) {
#pragma unroll
            for (int b = 0; b < OBI; ++b) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    if (ch < L.C) out_row[4 + ch] = io[b][r];
                }
            }
        }
    }
    DMN_STAMP(4);
    // cat[rgb, density, ins]  (dm_nerf.py:105)
    const int64_t ms = sample_of_lane();
    float* __restrict__ out_row = a.raw + (ms < a.M ? ms : a.M - 1) * (4 + L.C);
    if (ms < a.M && fresh(half) == 0) {
        out_row[0] = rgb_out[0];
        out_row[1] = rgb_out[1];
        out_row[2] = rgb_out[2];
        out_row[3] = sigma;
    }
    DMN_STAMP(5);
}

template <bool EMBEDDED, bool SAVE, bool FUSED = false>
int launch(const MlpArgs& a, hipStream_t stream) {
    const int64_t nblk = (a.M + 31) / 32;
    const int64_t grid = (nblk + 3) / 4;
    if (grid > 0x7fffffffLL) return dmn_fail(DMNERF_E_ARG, "mlp_fwd: %lld samples is too many for one launch", (long long)a.M);
    if (SAVE && a.M > DMNERF_MAX_TRAIN_SAMPLES)
        return dmn_fail(DMNERF_E_ARG, "mlp_fwd_train: %lld samples per launch exceed %lld (32-bit row offsets); split the batch",
                        (long long)a.M, (long long)DMNERF_MAX_TRAIN_SAMPLES);
    dim3 g((unsigned)grid), b(256);
    constexpr size_t lds_bytes = (size_t)(LDS_FLOATS + PARK_FLOATS) * sizeof(float);   // 163 840 B = the CU's whole LDS: one workgroup per CU
#define DMN_LAUNCH(OBI_)                                                                                              \
    {                                                                                                                \
        static DmnOncePerDevice once;                                                                                 \
        if (hipError_t e_ = once.run([] { return hipFuncSetAttribute((const void*)(mlp_fwd_kernel<OBI_, EMBEDDED, SAVE, FUSED>),              \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }); e_ != hipSuccess) \
            return dmn_fail_hip(e_, "mlp_fwd: hipFuncSetAttribute");                                       \
        hipLaunchKernelGGL((mlp_fwd_kernel<OBI_, EMBEDDED, SAVE, FUSED>), g, b, lds_bytes, stream, a);                       \
    }
    switch (a.L.OBI) {
        case 1: DMN_LAUNCH(1) break;
        case 2: DMN_LAUNCH(2) break;
        case 3: DMN_LAUNCH(3) break;
        case 4: DMN_LAUNCH(4) break;
        default: return dmn_fail(DMNERF_E_ARG, "mlp_fwd: unsupported logit count C=%d", a.L.C);
    }
#undef DMN_LAUNCH
    return dmn_check_launch("mlp_fwd");
}

}  // namespace
