// mlp_bwd_split.hip -- OPT-IN data-gradient kernel on the split-bf16 MFMA path (args.mfma_split in training).
//
// The dgrad of mlp_bwd.hip (31 weight quarters: ins_linear^T | F^T | mlps.7^T .. mlps.1^T, re-associated heads) with the
// GEMM engine of the split inference kernel (mlp_split_impl.h::gemm_split): W^T pre-split into three bf16 planes
// (layout.h::SplitTLayout), the gradient that a stage consumes split on the fly into three planes of packed bf16 pairs, six
// bf16 products per f32 product accumulated in f32 -- f32-class results at 2.7x fewer MFMA cycles.  What it reads
// (bit masks, dL/draw) and writes (dy rows in the f32 SaveLayout workspace, d raw transposed) is exactly what mlp_bwd.hip
// reads and writes, so the f32 weight-gradient kernel follows unchanged.  The ReLU-mask application, the f32 row store and
// the plane split of a dy are fused into one pass over the accumulators (there is no room for an f32 copy next to the
// planes).  Not the bitwise f32 chain of the default kernel, hence opt-in.
#include "mlp_split_impl.h"

namespace {

struct BwdSArgs {
    const float* blob;     // [table TAB_T_FLOATS f32 | split W^T stream]
    BlobLayout L;
    BlobTLayout LT;        // table offsets (w_rgbo, w_den)
    SplitTLayout S;
    const float* save;     // forward workspace (masks)
    const float* graw;     // [M, 4+C]
    float* dsave;          // gradients, same SaveLayout
    float* graw_t;         // d raw block-major [blk][4+C][32]
    int64_t M;
};

// d = acc . relu'(mask) per element -> stored as f32 rows (TID-addressed) and, if SPLIT, into the three bf16 planes P
template <int NB, bool SPLIT, int NW>
__device__ __forceinline__ void mask_store_split(const f32x16 (&acc)[NB], const unsigned (&m)[NB / 2], const RowIO& io, unsigned (&P)[3][NW]) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float x[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int p = 16 * b + r + e;
                const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)m[p >> 5], 31 - (p & 31), 1);
                x[e] = __uint_as_float(__float_as_uint(acc[b][r + e]) & keep);
                DMN_ACT_STORE_B32(f2u(x[e]), io.rs, run_off(0, r + e), (int)(io.soff + b * 4096), DMN_STORE_AUX);
            }
            if constexpr (SPLIT) {
                const int w = (2 * b + (r >> 3)) * 4 + ((r & 7) >> 1);
                split_pair(x[0], x[1], P[0][w], P[1][w], P[2][w]);
            }
            if ((r & 7) == 6) __builtin_amdgcn_sched_barrier(0);       // bound the live ranges: 8 elements at a time
        }
}

template <int NB>
__device__ __forceinline__ void zero_acc(f32x16 (&v)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) v[b] = (f32x16)(0.f);
}

template <int OBI>
__global__ __launch_bounds__(256) void mlp_bwd_split_kernel(const BwdSArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [ring 3 x 48 KiB][table 4 KiB]
    float* const tab = lds + SP_RING_FLOATS;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nblk = (a.M + 31) / 32;
    const int64_t blk_raw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t blk = blk_raw < nblk ? blk_raw : nblk - 1;
    const int64_t m_raw = blk * 32 + (lane & 31);
    const bool valid = m_raw < a.M;
    const int64_t m = valid ? m_raw : a.M - 1;
    const BlobLayout& L = a.L;
    const BlobTLayout& LT = a.LT;
    const SaveLayout SL = make_save_layout(a.M);
    const int64_t MP = save_row_len(a.M);

    // ---- incoming gradient (tail lanes: zero), d raw transposed, bit masks, table: as mlp_bwd.hip
    const float* __restrict__ gr = a.graw + m * (4 + L.C);
    const float g_rgb[3] = {valid ? gr[0] : 0.f, valid ? gr[1] : 0.f, valid ? gr[2] : 0.f};
    const float g_sigma = valid ? gr[3] : 0.f;
    const int GR = 4 + L.C;
    rsrc_t grs = uniform_rsrc(a.graw_t, a.graw_t ? (int64_t)GR * MP : 0);
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
            __builtin_amdgcn_raw_buffer_store_b32(f2u(gi[b][r]), grs, in ? gv + (4 + ch) * 128 : 0x7ffffff0, 0, 0);
        }
    if (half == 0) {
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_rgb[0]), grs, gv, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_rgb[1]), grs, gv + 128, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_rgb[2]), grs, gv + 256, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_sigma), grs, gv + 384, 0, 0);
    }
    // bit masks: fetched where they are used (4 words per lane and layer; no room to hold all eight next to the planes)
    const unsigned* const bw = reinterpret_cast<const unsigned*>(a.save + SL.bits) + blk * BITS_WORDS_PER_BLOCK;
    auto hmask = [&](int l, unsigned (&mb)[4]) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(bw + l * 256 + lane * 4);
        mb[0] = v[0]; mb[1] = v[1]; mb[2] = v[2]; mb[3] = v[3];
    };
    unsigned g1bits[2], g2bits[2];
    g1bits[0] = bw[2048 + lane * 2]; g1bits[1] = bw[2048 + lane * 2 + 1];
    g2bits[0] = bw[2176 + lane * 2]; g2bits[1] = bw[2176 + lane * 2 + 1];
    reinterpret_cast<f32x4*>(tab)[threadIdx.x] = (reinterpret_cast<const f32x4*>(a.blob) + threadIdx.x)[0];   // TAB_T_FLOATS = 256 x float4

    SStream ws;
    ws.rs = uniform_rsrc(a.blob, a.S.total);
    ws.wave = wave;
    ws.voff = (unsigned)(lane * 16 + wave * 1024);
    ws.ring = lds;
    ws.off = __builtin_amdgcn_readfirstlane((unsigned)(a.S.stream * 4));
    ws.cslot = 0;
#pragma unroll
    for (int i = 0; i < SP_DMA; ++i) ss_fetch_piece(ws, 0, i);
    ws.off += SP_SLOT_BYTES;
#pragma unroll
    for (int i = 0; i < SP_DMA; ++i) ss_fetch_piece(ws, 1, i);
    ws.off += SP_SLOT_BYTES;
    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");          // slot 0 landed, table visible
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const unsigned s0 = lds_addr(ws.ring) + lane * 16;
        static_for<4>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_read16_async<i * 1024>(ws.pre[i], s0); });
        lds_wait<0>(ws.pre);
    }

    f32x16 acc[8];
    unsigned Pd[3][64];                            // the dy a stage consumes, as three planes of bf16 pairs
    {
        // ---- heads: dg2 = relu'(g2) . (W_io^T g_ins) (nothing flows to h_7: h.detach()); dg1 on the VALU; dh_7 = F^T dg1
        f32x16 t4[4];
        unsigned Pn[3][32];                        // no consumer for dg2's planes (SPLIT = false below): scratch name only
        {
            unsigned Pg[3][OBI * 8];
            split_blocks<OBI, false>(gi, Pg);
            zero_acc<4>(t4);
            gemm_split<0, 2 * OBI, 4, 8>(ws, Pg, t4, lane);
        }
        mask_store_split<4, false>(t4, g2bits, make_rowio(a.dsave + SL.g1, 256, MP, blk, lane, 4), Pn);    // (dg1 | dg2: one 256-row tensor, layout.h)
        zero_acc<4>(t4);
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
        mask_store_split<4, true>(t4, g1bits, make_rowio(a.dsave + SL.g1, 256, MP, blk, lane), Pn);
        zero_acc<8>(acc);
        gemm_split<0, 8, 8, 8>(ws, Pn, acc, lane);
    }
    {
        // density_linear: dh_7 += w_d g_sigma;  dy_7 = dh_7 . relu'(h_7) -> rows + planes
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
        unsigned mb[4];
        hmask(7, mb);
        mask_store_split<8, true>(acc, mb, make_rowio(a.dsave + SL.h + (int64_t)7 * 256 * MP, 256, MP, blk, lane), Pd);
    }
    // ---- trunk: dh_{7-k} = W_{8-k}^T dy_{8-k}, k = 1..7
#pragma nounroll
    for (int st = 1; st <= NSTAGE_T; ++st) {
        unsigned mb[4];
        hmask(7 - st, mb);
        zero_acc<8>(acc);
        gemm_split<0, 16, 8, 8>(ws, Pd, acc, lane);          // (the last stage prefetches from the landing slots)
        const RowIO dio = make_rowio(a.dsave + SL.h + (int64_t)(7 - st) * 256 * MP, 256, MP, blk, lane);
        mask_store_split<8, true>(acc, mb, dio, Pd);                       // (dy_0's planes are not consumed: one loop body)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the last (landing-zone) fetches
}

}  // namespace

extern "C" int dmnerf_mlp_bwd_data_split(const float* d_blob_t_split, int ins_num, const float* d_save, const float* d_graw, int64_t M,
                                         float* d_dsave, float* d_graw_t, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data_split: ins_num %d unsupported", ins_num);
    if (M < 0 || M > DMNERF_MAX_TRAIN_SAMPLES) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data_split: M=%lld outside [0,%lld]", (long long)M, (long long)DMNERF_MAX_TRAIN_SAMPLES);
    if (M == 0) return DMNERF_OK;
    if (!d_blob_t_split || !d_save || !d_graw || !d_dsave) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data_split: null pointer");
    BwdSArgs a{};
    a.blob = d_blob_t_split; a.L = make_layout(ins_num); a.LT = make_layout_t(ins_num); a.S = make_split_layout_t(ins_num);
    a.save = d_save; a.graw = d_graw; a.dsave = d_dsave; a.graw_t = d_graw_t; a.M = M;
    const int64_t nblk = (M + 31) / 32;
    dim3 g((unsigned)((nblk + 3) / 4)), b(256);
    constexpr size_t lds_bytes = (size_t)(SP_RING_FLOATS + TAB_T_FLOATS) * sizeof(float);
#define DMN_LAUNCH(OBI_)                                                                                          \
    {                                                                                                            \
        static DmnOncePerDevice once;                                                                                 \
        if (hipError_t e_ = once.run([] { return hipFuncSetAttribute((const void*)mlp_bwd_split_kernel<OBI_>,              \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }); e_ != hipSuccess) \
            return dmn_fail_hip(e_, "mlp_bwd_data_split: hipFuncSetAttribute");                                       \
        hipLaunchKernelGGL(mlp_bwd_split_kernel<OBI_>, g, b, lds_bytes, (hipStream_t)stream, a);                        \
    }
    switch (a.L.OBI) {
        case 1: DMN_LAUNCH(1) break;
        case 2: DMN_LAUNCH(2) break;
        case 3: DMN_LAUNCH(3) break;
        case 4: DMN_LAUNCH(4) break;
        default: return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data_split: unsupported logit count C=%d", a.L.C);
    }
#undef DMN_LAUNCH
    return dmn_check_launch("mlp_bwd_data_split");
}
