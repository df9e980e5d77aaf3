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
// 7744 MFMAs per 32 samples at C=14: the reference's dgrad is fwd MACs - 101 248 = 592 256 MAC per sample (SURVEY.md 8
// a-12); folding the activation-free feature linears (layout.h, BlobTLayout) leaves 494 080 + the 1 / 3-row VALU heads.
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
    float* graw_t;         // optional: d raw in block-major form [blk][4+C][32] for the weight-gradient kernel
    int64_t M;
#ifdef DMN_FWD_TRACE
    long long* trace;      // diagnostic builds only (make diag): per-workgroup cycle stamps, see scripts/diag_fwd.py
#endif
};
#ifdef DMN_FWD_TRACE
static long long* g_bwd_trace = nullptr;
extern "C" int dmnerf_debug_bwd_trace(long long* p) { g_bwd_trace = p; return 0; }
#define DMN_STAMP(k) do { if (a.trace && threadIdx.x == 0) a.trace[8 * blockIdx.x + (k)] = (long long)clock64(); } while (0)
#else
#define DMN_STAMP(k) do {} while (0)
#endif

template <int NB>
__device__ __forceinline__ void zero(f32x16 (&v)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = (f32x16)(0.f);
}

// Same LDS weight-streaming engine as the forward (mlp_common.h): the W^T stream is consumed one 64 KiB
// quarter at a time, fetched one quarter ahead; the dy stores of a stage ride in the MFMA gaps of quarters
// 1..3 of the NEXT stage (while dy is its B operand), after that quarter's DMA pieces.
template <int OBI>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(const BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [ring 2 x 64 KiB][table 4 KiB]
    float* const tab = lds + RING_FLOATS;
    DMN_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nblk = (a.M + 31) / 32;
    const int64_t blk_raw = (int64_t)blockIdx.x * 4 + wave;
    const bool wave_active = blk_raw < nblk;
    const int64_t blk = wave_active ? blk_raw : nblk - 1;
    const int64_t m_raw = blk * 32 + (lane & 31);
    const bool valid = m_raw < a.M;          // (a wave beyond the batch duplicates the last block exactly)
    const int64_t m = m_raw < a.M ? m_raw : a.M - 1;

    const BlobLayout& L = a.L;
    const BlobTLayout& LT = a.LT;
    const SaveLayout SL = make_save_layout(a.M);
    const int64_t MP = save_row_len(a.M);
    const int srows = 1;                             // (every wave stores: see mlp_common.h::RowIO)

    // ---- oldest VMEM ops: incoming gradient, ReLU bit masks, table --------------------------------
    const float* __restrict__ gr = a.graw + m * (4 + L.C);
    // tail lanes get a zero gradient: every dy they produce is then exactly 0, so the padding
    // columns contribute nothing to the weight / bias gradients
    const float g_rgb[3] = {valid ? gr[0] : 0.f, valid ? gr[1] : 0.f, valid ? gr[2] : 0.f};
    const float g_sigma = valid ? gr[3] : 0.f;
    // Branch-free: a channel beyond C is loaded from the (valid) last channel and replaced by 0 with a select -- per-
    // element `if`s cost an exec-mask save each and the 48 / 64 of them of the Replica-width kernels spilled 37 / 67 SGPRs.
    // d raw is also written transposed for the weight-gradient kernel, [blk][4+C rows][32], element by element as it is
    // loaded (one live lane mask at a time).  No transposed output wanted: an empty descriptor drops every store.
    const int GR = 4 + L.C;
    rsrc_t grs = uniform_rsrc(a.graw_t, a.graw_t ? (int64_t)srows * GR * MP : 0);
    const int gv = (int)((blk * GR * 32 + (lane & 31)) * 4);
    f32x16 gi[OBI];
#pragma unroll
    for (int b = 0; b < OBI; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool in = ch < L.C;
            const float v = gr[4 + (in ? ch : L.C - 1)];
            gi[b][r] = (in && valid) ? v : 0.f;
            // rows beyond C do not exist: those lanes get an offset outside the descriptor's range and the hardware drops
            // the store (raw buffer bounds check) -- no branch
            __builtin_amdgcn_raw_buffer_store_b32(f2u(gi[b][r]), grs, in ? gv + (4 + ch) * 128 : 0x7ffffff0, 0, 0);
        }
    unsigned hbits[8][4], g1bits[2], g2bits[2];
    {
        const unsigned* bw = reinterpret_cast<const unsigned*>(a.save + SL.bits) + blk * BITS_WORDS_PER_BLOCK;
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(bw + l * 256 + lane * 4);
            hbits[l][0] = v[0]; hbits[l][1] = v[1]; hbits[l][2] = v[2]; hbits[l][3] = v[3];
        }
        g1bits[0] = bw[2048 + lane * 2]; g1bits[1] = bw[2048 + lane * 2 + 1];
        g2bits[0] = bw[2176 + lane * 2]; g2bits[1] = bw[2176 + lane * 2 + 1];
    }
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.blobT) + threadIdx.x;
        reinterpret_cast<f32x4*>(tab)[threadIdx.x] = src[0];              // TAB_T_FLOATS = 1024 = 256 x float4
    }
    if (half == 0) {
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_rgb[0]), grs, gv, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_rgb[1]), grs, gv + 128, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_rgb[2]), grs, gv + 256, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_sigma), grs, gv + 384, 0, 0);
    }
    WStream ws;
    ws_init(ws, a.blobT, LT.total, lds, lane, wave, LT.stream);
    ws_fetch_first(ws);                                                   // quarter 0: ins_linear^T

    DMN_STAMP(1);
    f32x16 d[8], acc[8];
    {
        // ---- heads.  ins branch: dg2 = relu'(g2) . (W_io^T g_ins); it sends nothing to h_7 (h.detach(), dm_nerf.py:95) and
        // d ins_feature is never formed (its weight gradients come from Q = dg2 . h_7^T, heads.hip)
        f32x16 t4[4], dg2[4], dg1[4];
        ws_prime<4>(ws, lane);
        gemm_quarter<0, 4 * OBI, 4, 8, true>(ws, gi, t4, lane);
        apply_mask<4>(dg2, g2bits, t4);

        DMN_STAMP(2);
        // rgb branch: dg1 = relu'(g1) . (W_ro^T g_rgb) on the VALU
        zero<4>(t4);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4* wr = reinterpret_cast<const f32x4*>(tab + LT.w_rgbo + (c * 2 + half) * 64);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 w = wr[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = 4 * i + j;
                    t4[p >> 4][p & 15] = fmaf(w[j], g_rgb[c], t4[p >> 4][p & 15]);
                }
            }
        }
        apply_mask<4>(dg1, g1bits, t4);
        store_rows<4>(make_rowio(a.dsave + SL.g1, 256, srows * MP, blk, lane), dg1);          // (dg1 | dg2: one 256-row tensor, layout.h)      // burst (once per block)
        // d h_7 (rgb branch) = F^T dg1, F = rgb_hidden[:, :256] . rgb_feature (layout.h): one 128 -> 256 GEMM, two
        // quarters; dg2 is saved meanwhile (32 + 32 spread stores)
        const RowIO g2io = make_rowio(a.dsave + SL.g1, 256, srows * MP, blk, lane, 4);
        auto st_g2 = [&](int k0) { return [&, k0](int k) { store_row_one(g2io, dg2, k0 + k); }; };
        gemm_quarter<0, 8, 8, 8, true, 32>(ws, dg1, acc, lane, st_g2(0));
        gemm_quarter<8, 8, 8, 8, false, 32>(ws, dg1, acc, lane, st_g2(32));
    }
    {
        // density_linear (dm_nerf.py:101): dh_7 += w_d * g_sigma;  dy_7 = dh_7 . relu'(h_7)
        const f32x4* wd = reinterpret_cast<const f32x4*>(tab + LT.w_den + half * 128);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const f32x4 w = wd[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = 4 * i + j;
                acc[p >> 4][p & 15] = fmaf(w[j], g_sigma, acc[p >> 4][p & 15]);
            }
        }
        apply_mask<8>(d, hbits[7], acc);
    }

    DMN_STAMP(3);
    // ---- trunk: st = k: dh_{7-k} = W_{8-k}^T dy_{8-k}, k = 1..7.  The stage's input d (dy_7 .. dy_1) is saved while it
    // is consumed: 43 + 43 + 42 TID-addressed stores in the MFMA gaps of quarters 1..3.
#pragma nounroll
    for (int st = 1; st <= NSTAGE_T; ++st) {
        const RowIO dio = make_rowio(a.dsave + SL.h + (int64_t)(8 - st) * 256 * MP, 256, srows * MP, blk, lane);
        auto st_d = [&](int k0) { return [&, k0](int k) { store_row_one(dio, d, k0 + k); }; };
        gemm_quarter<0, 8, 8, 8, true>(ws, d, acc, lane);
        gemm_quarter<8, 8, 8, 8, false, 43>(ws, d, acc, lane, st_d(0));
        gemm_quarter<16, 8, 8, 8, false, 43>(ws, d, acc, lane, st_d(43));
        gemm_quarter<24, 8, 8, 8, false, 42>(ws, d, acc, lane, st_d(86));   // (the last stage prefetches from the landing zone)
        // dy_l = dh_l . relu'(h_l), l = 7 - st: the bit masks were loaded up front; select the layer with a
        // wave-uniform switch (register arrays cannot be indexed dynamically)
        unsigned mb[4];
        switch (7 - st) {
#define DMN_PICK(l_) case l_: mb[0] = hbits[l_][0]; mb[1] = hbits[l_][1]; mb[2] = hbits[l_][2]; mb[3] = hbits[l_][3]; break;
            DMN_PICK(0) DMN_PICK(1) DMN_PICK(2) DMN_PICK(3) DMN_PICK(4) DMN_PICK(5) default: DMN_PICK(6)
#undef DMN_PICK
        }
        apply_mask<8>(d, mb, acc);
    }
    DMN_STAMP(4);
    store_rows<8>(make_rowio(a.dsave + SL.h, 256, srows * MP, blk, lane), d);             // dy_0
    DMN_STAMP(5);
}

}  // namespace

extern "C" int dmnerf_mlp_bwd_data(const float* d_blob, const float* d_blob_t, int ins_num, const float* d_save,
                                   const float* d_graw, int64_t M, float* d_dsave, float* d_graw_t, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data: ins_num %d unsupported", ins_num);
    if (M < 0 || M > DMNERF_MAX_TRAIN_SAMPLES) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data: M=%lld outside [0,%lld]", (long long)M, (long long)DMNERF_MAX_TRAIN_SAMPLES);
    if (M == 0) return DMNERF_OK;
    if (!d_blob || !d_blob_t || !d_save || !d_graw || !d_dsave) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data: null pointer");
    BwdArgs a{};
    a.blob = d_blob; a.blobT = d_blob_t; a.L = make_layout(ins_num); a.LT = make_layout_t(ins_num);
    a.save = d_save; a.graw = d_graw; a.dsave = d_dsave; a.graw_t = d_graw_t; a.M = M;
#ifdef DMN_FWD_TRACE
    a.trace = g_bwd_trace;
#endif
    const int64_t nblk = (M + 31) / 32;
    dim3 g((unsigned)((nblk + 3) / 4)), b(256);
    constexpr size_t lds_bytes = (size_t)(RING_FLOATS + TAB_T_FLOATS) * sizeof(float);
#define DMN_LAUNCH(OBI_)                                                                                          \
    {                                                                                                            \
        static DmnOncePerDevice once;                                                                                 \
        if (hipError_t e_ = once.run([] { return hipFuncSetAttribute((const void*)mlp_bwd_kernel<OBI_>,              \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }); e_ != hipSuccess) \
            return dmn_fail_hip(e_, "mlp_bwd_data: hipFuncSetAttribute");                                       \
        hipLaunchKernelGGL(mlp_bwd_kernel<OBI_>, g, b, lds_bytes, (hipStream_t)stream, a);                        \
    }
    switch (a.L.OBI) {
        case 1: DMN_LAUNCH(1) break;
        case 2: DMN_LAUNCH(2) break;
        case 3: DMN_LAUNCH(3) break;
        case 4: DMN_LAUNCH(4) break;
        default: return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data: unsupported logit count C=%d", a.L.C);
    }
#undef DMN_LAUNCH
    return dmn_check_launch("mlp_bwd_data");
}
