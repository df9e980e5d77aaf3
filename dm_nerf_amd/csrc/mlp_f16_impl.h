// mlp_f16_impl.h -- OPT-IN kernel (entry points: mlp_f16.hip inference, mlp_f16_train.hip training forward): the DM-NeRF MLP on
// v_mfma_f32_32x32x16_f16 with every f32 operand split into TWO f16 planes ("f16x2", split_f16.h): three MFMAs per f32 product
// instead of the six of the bf16x3 kernel (mlp_split_impl.h), f32-class results (NOT the bitwise fmaf chain of the default
// kernel: opt-in, args.mfma_split = "f16x2").  It evaluates the fused-heads function (weights.py::fuse_heads).
//
// Structure.  As in the other fused kernels one wave owns 32 samples for the whole network and the accumulator layout of a
// layer is the B-operand layout of the next (a lane's 8 accumulator registers r = 8 t + q of out-block b are its 8 k-slots of
// k-block 2 b + t), so activations never leave registers: they are kept as two planes of packed f16 pairs.  What is new is
// the loop order: OUT-BLOCK-OUTER.  A "pass" accumulates NOB out-blocks (2 in the trunk) over ALL k-blocks of the layer, so
//   * only 2 x NOB accumulator blocks are live (one set accumulating, one set being post-processed) instead of the layer's 8,
//     which pays for a SECOND plane set: a layer reads planes A and writes planes B, the next reads B and writes A;
//   * the ReLU + bias + split epilogue of pass p - 1 (16 element pairs, ~9 VALU instructions each) rides in the MFMA gaps of
//     pass p -- a 32x32x16 MFMA hides ~5 single-issue instructions (MI355X guide) -- instead of standing between two layers
//     (in the k-outer bf16x3 kernel that burst is 6 % of the time; here it would be ~15 %, the MFMA time having halved);
//   * the last pass of a layer is post-processed under the first three groups of the next layer's first pass (its k-blocks
//     12..15 are not needed before the fourth group).
// Weights stream through an LDS ring of F16_RING = 8 group slots of 16 KiB (layout.h: a group = 8 hi tiles + 8 lo tiles = the A
// operands of 24 MFMAs) by MUBUF LDS-DMA F16_LA = 6 groups ahead, 4 pieces per wave and group; ring position and the hand-over
// parity are run-time (wave-uniform), the schedule inside a group is static:
//   gaps 0..7    hi tile j x x_hi      (issue the read of lo tile j; wait for hi tile j)
//   gaps 8..15   hi tile j x x_lo      (first group of a pass: the bias quads of the pass being post-processed)
//   gap 16       every second group: hand-over (all reads of this group returned, groups <= g + 2 landed, s_barrier)
//   gaps 16..23  lo tile j x x_hi      (issue the read of the NEXT group's hi tile j; wait for lo tile j)
// All LDS reads are inline asm (invisible to the compiler's waitcnt insertion) with hand-counted lgkmcnt waits (LDS returns in
// order); scripts/check_asm_hazard.py proves on the ISA that no register is touched before its read has been waited for.
// LDS: 8 x 16 KiB ring + 16 KiB bias table = 144 KiB.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "layout.h"
#include "mlp_common.h"
#include "split_f16.h"

using namespace dmn;

namespace {

constexpr int F16_RING_FLOATS = F16_RING * F16_GROUP_WORDS;      // 128 KiB
constexpr int F16_LDS_FLOATS = F16_RING_FLOATS + TAB_FLOATS;     // + 16 KiB table
constexpr int F16_GROUP_BYTES = F16_GROUP_WORDS * 4;
static_assert((F16_RING & (F16_RING - 1)) == 0 && F16_LA + 2 <= F16_RING, "ring / look-ahead");

struct F16Args {
    const float* blob;      // [bias table f32 | group stream]
    F16Layout S;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    float* raw;
    float* save;            // SAVE: the training workspace of layout.h::SaveLayout
    int64_t M;
    int Sr;                 // samples per ray
#ifdef DMN_F16_TRACE
    long long* trace;       // diagnostic builds only (make f16var FLAGS=-DDMN_F16_TRACE): per-workgroup cycle stamps, scripts/diag_f16.py
#endif
};
#ifdef DMN_F16_TRACE
#define DMN_F16_STAMP(k) do { if (a.trace && threadIdx.x == 0) { a.trace[8 * blockIdx.x + (k)] = (long long)clock64(); if ((k) == 0 || (k) == 5) a.trace[8 * blockIdx.x + 6 + ((k) == 5)] = (long long)wall_clock64(); } } while (0)
#else
#define DMN_F16_STAMP(k) do {} while (0)
#endif

struct GStream {
    rsrc_t rs;
    unsigned voff;          // lane*16 + wave*1024
    float* ring;
    int wave;
    unsigned off;           // byte offset (from the blob start) of group gidx + F16_LA, the next one to FETCH
    int gidx;               // global index of the group being consumed (wave-uniform)
    unsigned lane16;        // LDS byte address of this lane's 16 bytes of tile 0 of ring slot 0
    unsigned cur, nxt;      // ... of the slot being consumed / of the next group's slot
    f32x4 H[8], Lo[8];      // hi / lo tiles
};

template <int N>
__device__ __forceinline__ void wait_lgkm() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt");
    __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
__device__ __forceinline__ void hand_back(f32x4& v) { asm volatile("" : "+" DMN_TILE_RC(v)); }

// bias quads go to VGPRs (the VALU adds them)
template <int OFF>
__device__ __forceinline__ void lds_read16_v(f32x4& v, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_read_b128 offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

// piece i (0..3) of group gidx + F16_LA: wave w fetches the 1 KiB pieces 4 i + w of its 16 KiB
__device__ __forceinline__ void gs_fetch_piece(const GStream& ws, int slot_group, int i) {
#ifdef DMN_F16_NODMA
    return;                                                      // ablation (timing only): no weight traffic
#endif
    float* dst = ws.ring + (slot_group & (F16_RING - 1)) * F16_GROUP_WORDS + ws.wave * 256 + i * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ws.rs, (DMN_LAS void*)dst, 16, (int)ws.voff, (int)(ws.off + i * 4096), 0, 0);
}

struct NoSideC {
    template <int G>
    __device__ __forceinline__ void operator()(std::integral_constant<int, G>) const {}
};

// The epilogue of a finished pass as side operations of the following one(s): bias + ReLU (+ SAVE: f32 row store and 1-bit ReLU
// mask) + split of NOBP accumulator blocks (out-blocks OB0 ..) into the plane words of the next layer.  A 32x32x16 MFMA hides ~5
// single-issue instructions and every gap already carries a tile read and its wait, so a pair's instructions are dealt out
// over NPH consecutive gaps, pair k at gaps G0 + NPH k ..:
//   phase 0 : 2 v_accvgpr_read (the accumulators live in AGPRs, which the VALU cannot read) + 2 bias adds
//   phase 1 : 2 ReLUs (+ SAVE: the two TID-addressed row stores; else: the hi word)
//   SAVE 2  : mask bits (compare + add-with-carry per element) + the hi word
//   last    : the 2 residuals (v_fma_mix_f32) + the lo word
// (Measured: with a pair's instructions in ONE gap the trunk ran at 38.6 instead of 32 cycles per MFMA, profiles/r03/.)
// HASQ: bq[QB0 ..] holds the pass's bias quads, 4 per block.  G0 >= 3: the pass's last MFMA must have retired before the asm
// accumulator reads, which the compiler's hazard recognizer does not see.
struct SaveCtx {                                // SAVE: where a pass's rows and mask word go
    RowIO io;                                   // the layer's saved tensor (rows of this wave's 32-sample block)
    rsrc_t bits_rs;
    int bits_voff;                              // (blk * BITS_WORDS_PER_BLOCK + lane * words per lane) * 4
};
template <bool SAVE, int NOBP, int OB0, int QB0, bool HASQ, bool RELU, int G0, int NUM, int NWO, int NBQA>
struct EpiFwd {                                 // NUM pairs per burst of NPH gaps (1 where the carrying pass is long enough)
    static constexpr int NPH = SAVE ? 4 : 3;
    f32x16 (&Y)[NOBP];
    f32x4 (&bq)[NBQA];
    unsigned (&Ohi)[NWO];
    unsigned (&Olo)[NWO];
    const SaveCtx* sv;                          // SAVE only
    int bits_soff;                              // SAVE only: byte offset of the pass's first mask word behind bits_voff
    float xs0[NUM] = {}, xs1[NUM] = {};
    unsigned mw = 0u;
    template <int GAP>
    __device__ __forceinline__ void operator()(std::integral_constant<int, GAP>) {
        static_assert(G0 >= 3, "the finished pass's last MFMA must have retired");
        if constexpr (HASQ && GAP == G0) {
#pragma unroll
            for (int q = 0; q < 4 * NOBP; ++q) asm volatile("" : "+v"(bq[QB0 + q]));
        }
        if constexpr (GAP >= G0 && (GAP - G0) / NPH * NUM < 8 * NOBP)
            static_for<NUM>([&](auto nc) { one<(GAP - G0) % NPH, (GAP - G0) / NPH * NUM + decltype(nc)::value, decltype(nc)::value>(); });
    }
    template <int ph, int k, int n>
    __device__ __forceinline__ void one() {
        float& x0 = xs0[n];
        float& x1 = xs1[n];
        if constexpr (k < 8 * NOBP) {
            constexpr int b = k / 8, r = 2 * (k % 8);
            constexpr int w = (2 * (OB0 + b) + (r >> 3)) * 4 + ((r & 7) >> 1);
            static_assert(w < NWO, "plane word");
#ifdef DMN_F16_NOSIDE
            if (k > 0) return;                                   // ablation (timing only): one pair per pass keeps the data flow alive
#endif
            if constexpr (ph == 0) {
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(Y[b][r]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(Y[b][r + 1]));
                if constexpr (HASQ) {
                    x0 += bq[QB0 + b * 4 + (r >> 2)][r & 3];
                    x1 += bq[QB0 + b * 4 + ((r + 1) >> 2)][(r + 1) & 3];
                }
                asm volatile("" : "+v"(x0), "+v"(x1));           // the adds stay in this gap
            } else if constexpr (ph == 1) {
                if constexpr (RELU) { x0 = relu1(x0); x1 = relu1(x1); }
                asm volatile("" : "+v"(x0), "+v"(x1));
                if constexpr (SAVE) {
                    DMN_ACT_STORE_B32(f2u(x0), sv->io.rs, run_off(0, r), (int)(sv->io.soff + (OB0 + b) * 4096), DMN_STORE_AUX);
                    DMN_ACT_STORE_B32(f2u(x1), sv->io.rs, run_off(0, r + 1), (int)(sv->io.soff + (OB0 + b) * 4096), DMN_STORE_AUX);
                } else {
                    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(Ohi[w]) : "v"(x0), "v"(x1));
                }
            } else if constexpr (SAVE && ph == 2) {
                if constexpr ((k & 15) == 0) mw = 0u;
                asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                             "v_cmp_lt_f32 vcc, 0, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mw) : "v"(x0), "v"(x1) : "vcc");
                asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(Ohi[w]) : "v"(x0), "v"(x1));
                if constexpr ((k & 15) == 15)                    // 32 elements = one mask word per lane (element p = 32 w + i is bit 31 - i)
                    __builtin_amdgcn_raw_buffer_store_b32(mw, sv->bits_rs, sv->bits_voff, bits_soff + (k >> 4) * 4, 0);
            } else {
                float r0, r1;
                asm volatile("v_fma_mix_f32 %0, -%2, 1.0, %3 op_sel_hi:[1,0,0]\n\t"
                             "v_fma_mix_f32 %1, -%2, 1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                             : "=&v"(r0), "=&v"(r1) : "v"(Ohi[w]), "v"(x0), "v"(x1));
                asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(Olo[w]) : "v"(r0), "v"(r1));
            }
        }
    }
};

// One group = 24 MFMAs: tile i <-> (k-block KB0 + i / NOB of the planes, accumulator block i % NOB).
// ZEROC: first group of a pass (C = 0 for the first MFMA of every block).  NBQ: bias quads read at gaps 8..15 from baddr into
// bq[QB0 ..] (4 per block, blocks QSTR bytes apart: 128 in the bias table).  VMY: VMEM operations other than weight pieces (activation stores) that are guaranteed to be younger than the
// pieces of group gidx + 2 at the hand-over (a LOWER bound: fewer means waiting for a few of the oldest younger pieces too).
template <int NOB, int KB0, bool ZEROC, int NBQ, int QB0, int GAP0, int VMY, int QSTR, int NW, int NBQA, class Side>
__device__ __forceinline__ void f16_group(GStream& ws, const unsigned (&Phi)[NW], const unsigned (&Plo)[NW], f32x16 (&acc)[NOB],
                                          f32x4 (&bq)[NBQA], unsigned baddr, Side&& side) {
    static_assert(8 % NOB == 0 && (KB0 + 8 / NOB) * 4 <= NW, "B planes too small");
    static_assert(NBQ <= 16 && QB0 + NBQ <= NBQA, "bias quads");
    static_assert(4 * (F16_LA - 2) + VMY <= 63, "vmcnt");
    constexpr int BPG = (NBQ + 7) / 8;                           // bias quads per gap
    constexpr int LGK_LO = 8 + NBQ > 15 ? 15 : 8 + NBQ;          // reads younger than lo tile j at gap 16 + j (clamped: waits for more)
    // ---- hi tiles: x_hi then x_lo
    static_for<16>([&](auto ic) {
        constexpr int i = decltype(ic)::value, j = i & 7, t = i >> 3, ob = j % NOB, kb = KB0 + j / NOB;
        if constexpr (t == 0) {
            lds_read16_async<8192 + j * 1024>(ws.Lo[j], ws.cur);
            wait_lgkm<8>();                                      // younger than hi tile j: hi j+1..7, lo 0..j
            hand_back(ws.H[j]);
        } else {
            static_for<BPG>([&](auto sc) {
                constexpr int q = j * BPG + decltype(sc)::value;
                if constexpr (q < NBQ) lds_read16_v<(q >> 2) * QSTR + (q & 3) * 16>(bq[QB0 + q], baddr);
            });
        }
        if constexpr ((i & 3) == 2) gs_fetch_piece(ws, ws.gidx + F16_LA, i >> 2);
        side(std::integral_constant<int, GAP0 + i>{});
        if constexpr (ZEROC && t == 0 && j < NOB)
            acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_ah(ws.H[j]), as_bh(&Phi[kb * 4]), (f32x16)(0.f), 0, 0, 0);
        else if constexpr (t == 0)
            acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_ah(ws.H[j]), as_bh(&Phi[kb * 4]), acc[ob], 0, 0, 0);
        else
            acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_ah(ws.H[j]), as_bh(&Plo[kb * 4]), acc[ob], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    });
    // ---- hand-over, every second group: every read of this slot (and the previous one) by this wave has returned -- the NBQ
    // younger table reads may still fly --, the pieces of groups <= gidx + 2 have landed in every wave's view (vmcnt retires in
    // order: the 4 (F16_LA - 2) youngest pieces belong to later groups), and the barrier publishes both facts.
    if (ws.gidx & 1) {
        wait_lgkm<(NBQ > 15 ? 15 : NBQ)>();
        wait_vm<4 * (F16_LA - 2) + VMY>();
        asm volatile("" ::: "memory");
#ifndef DMN_F16_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- lo tiles x x_hi; the next group's hi tiles are read meanwhile
    static_for<8>([&](auto jc) {
        constexpr int j = decltype(jc)::value, ob = j % NOB, kb = KB0 + j / NOB;
        lds_read16_async<j * 1024>(ws.H[j], ws.nxt);
        wait_lgkm<LGK_LO>();                                     // younger than lo tile j: lo j+1..7, the bias quads, next hi 0..j
        hand_back(ws.Lo[j]);
        side(std::integral_constant<int, GAP0 + 16 + j>{});
        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_ah(ws.Lo[j]), as_bh(&Phi[kb * 4]), acc[ob], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    });
    ws.off += F16_GROUP_BYTES;
    ws.gidx += 1;
    ws.cur = ws.nxt;
    ws.nxt = ws.lane16 + (unsigned)(((ws.gidx + 1) & (F16_RING - 1)) * F16_GROUP_BYTES);
}

// NG consecutive groups on the same accumulator blocks: k-blocks KB0 .. of the planes, 8 / NOB per group; the first one reads
// NBQ bias quads into bq[QB0 ..]
template <int NOB, int NG, int KB0, int NBQ, int QB0, int GAP0, bool ZERO_FIRST, int VMY, int QSTR = 128, int NW, int NBQA, class Side>
__device__ __forceinline__ void f16_pass(GStream& ws, const unsigned (&Phi)[NW], const unsigned (&Plo)[NW], f32x16 (&acc)[NOB],
                                         f32x4 (&bq)[NBQA], unsigned baddr, Side&& side) {
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        f16_group<NOB, KB0 + g * (8 / NOB), (ZERO_FIRST && g == 0), (g == 0 ? NBQ : 0), QB0, GAP0 + 24 * g, VMY, QSTR>(ws, Phi, Plo, acc, bq, baddr, side);
    });
}

// planes of NV accumulator-layout f32x16 blocks (the encodings): block b, register r = 8 t + q -> k-block 2 b + t, word (q >> 1)
template <int NV>
__device__ __forceinline__ void split_blocks_f16(const f32x16 (&x)[NV], unsigned (&Phi)[NV * 8], unsigned (&Plo)[NV * 8]) {
#pragma unroll
    for (int b = 0; b < NV; ++b)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int w = (2 * b + (r >> 3)) * 4 + ((r & 7) >> 1);
            split_pair_f16(x[b][r], x[b][r + 1], Phi[w], Plo[w]);
        }
}

template <int OBX, bool SAVE = false>
__global__ __launch_bounds__(256) void mlp_f16_kernel(const F16Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [ring 8 x 16 KiB][table 16 KiB]
    float* const tab = lds + F16_RING_FLOATS;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nblk = (a.M + 31) / 32;
    const int64_t blk_raw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t blk = blk_raw < nblk ? blk_raw : nblk - 1;             // (a wave beyond the batch duplicates the last block)
    const int64_t m_raw = blk * 32 + (lane & 31);
    const bool valid = m_raw < a.M;
    const int64_t m = valid ? m_raw : a.M - 1;
    const int C = a.S.C;

    DMN_F16_STAMP(0);
    float pt[3], vd[3];
    {
        const int64_t n = m / a.Sr;
        const float ox = a.rays_o[n * 3 + 0], oy = a.rays_o[n * 3 + 1], oz = a.rays_o[n * 3 + 2];
        const float dx = a.rays_d[n * 3 + 0], dy = a.rays_d[n * 3 + 1], dz = a.rays_d[n * 3 + 2];
        const float zv = a.z[m];
        pt[0] = ox + dx * zv; pt[1] = oy + dy * zv; pt[2] = oz + dz * zv;          // render.py:49
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        vd[0] = dx / nrm; vd[1] = dy / nrm; vd[2] = dz / nrm;                       // render.py:37
    }
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.blob) + threadIdx.x;
        f32x4* dst = reinterpret_cast<f32x4*>(tab) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < TAB_FLOATS / 1024; ++k) dst[k * 256] = src[k * 256];
    }
    GStream ws;
    ws.rs = uniform_rsrc(a.blob, a.S.total);
    ws.wave = wave;
    ws.voff = (unsigned)(lane * 16 + wave * 1024);
    ws.ring = lds;
    ws.off = __builtin_amdgcn_readfirstlane((unsigned)(a.S.stream * 4));
    ws.gidx = 0;
    ws.lane16 = lds_addr(lds) + lane * 16;
    // prologue: groups 0 .. F16_LA - 1 into ring slots 0 .. F16_LA - 1
#pragma unroll
    for (int g = 0; g < F16_LA; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gs_fetch_piece(ws, g, i);
        ws.off += F16_GROUP_BYTES;
    }                                                                    // from now on `off` = group gidx + F16_LA

    // SAVE (opt-in training forward): the f32 workspace of layout.h::SaveLayout that the backward kernels consume -- pe, de,
    // the ReLU outputs h_0 .. h_7, g1, g2 as block-major rows (TID-addressed stores, mlp_common.h::RowIO) and the 1-bit masks
    const SaveLayout SL = make_save_layout(a.M);
    const int64_t MP = save_row_len(a.M);
    rsrc_t bits_rs = uniform_rsrc(SAVE ? a.save + SL.bits : a.blob, SAVE ? (int64_t)(BITS_WORDS_PER_BLOCK / 32) * MP : 0);
    auto save_ctx = [&](int64_t tensor_off, int rows, int words_per_lane, int word0) -> SaveCtx {
        SaveCtx c;
        c.io.rs = bits_rs; c.io.soff = 0u; c.bits_rs = bits_rs; c.bits_voff = 0;      // (inference: never used)
        if constexpr (SAVE) {
            c.io = make_rowio(a.save + tensor_off, rows, MP, blk, lane);
            c.bits_rs = bits_rs;
            c.bits_voff = (int)((blk * BITS_WORDS_PER_BLOCK + word0 + lane * words_per_lane) * 4);
        }
        return c;
    };

    // encodings as planes.  The pad slot of the position encoding (k-pair 1, upper half) carries 1.0: the stream holds the bias
    // of mlps.0 in that column (pack.cpp), so the first layer needs no bias table
    unsigned Ppe[2][16], Pde[2][8];
    {
        f32x16 pe[2], de[1];
        encode<POS_L, 2>(pt, pe, half);
        encode<DIR_L, 1>(vd, de, half);
        if constexpr (SAVE) {
            store_encoded_rows<POS_L, 2>(a.save + SL.pe, MP, blk, lane, pe);
            store_encoded_rows<DIR_L, 1>(a.save + SL.de, MP, blk, lane, de);
        }
        pe[0][1] = half ? 1.f : pe[0][1];
        split_blocks_f16<2>(pe, Ppe[0], Ppe[1]);
        split_blocks_f16<1>(de, Pde[0], Pde[1]);
    }

    // groups 0 and 1 landed (the pieces of groups 2 .. 5 -- and, SAVE, the 90 younger encoding stores -- may still fly), table
    // visible; hi tiles of group 0
    if constexpr (SAVE) wait_vm<63>(); else wait_vm<4 * (F16_LA - 2)>();
    wait_lgkm<0>();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ws.cur = ws.lane16;
    ws.nxt = ws.lane16 + F16_GROUP_BYTES;
    static_for<8>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_read16_async<i * 1024>(ws.H[i], ws.cur); });

    DMN_F16_STAMP(1);
    f32x16 acc0[2], acc1[2];                   // the two accumulator sets of the trunk passes (even / odd pass)
    unsigned PA[2][64], PB[2][64];             // the two plane sets: 256 features as (hi, lo) words of f16 pairs
    f32x4 bq[16];                              // bias quads: trunk pass p reads ITS OWN 8 into half p & 1 (used one pass later)
    // bias table walk (this lane's half): mlps.1's first pass, then pass by pass
    unsigned baddr = lds_addr(tab) + half * 64 + 256 * 4;
    // activation stores guaranteed younger than the awaited weight pieces at a trunk hand-over: the window back to their issue is
    // 98 gaps, a pass issues 33 stores in gaps 3..66 of its 96 (mlps.5: 120) -- at least 20 fall into any window
    constexpr int VMT = SAVE ? 16 : 0;
    constexpr int NU0 = SAVE ? 4 : 3;          // mlps.0's one-group passes carry their predecessor's epilogue NU0 pairs at a time

    // ---- mlps.0 : 63 -> 256, four passes of one group; the epilogue of pass p - 1 rides in pass p
    const SaveCtx sv0 = save_ctx(SL.h, 256, 4, 0);
    f16_pass<2, 1, 0, 0, 0, 0, true, 0>(ws, Ppe[0], Ppe[1], acc0, bq, baddr, NoSideC{});
    f16_pass<2, 1, 0, 0, 0, 0, true, 0>(ws, Ppe[0], Ppe[1], acc1, bq, baddr, EpiFwd<SAVE, 2, 0, 0, false, true, 3, NU0, 64, 16>{acc0, bq, PA[0], PA[1], &sv0, 0});
    f16_pass<2, 1, 0, 0, 0, 0, true, 0>(ws, Ppe[0], Ppe[1], acc0, bq, baddr, EpiFwd<SAVE, 2, 2, 0, false, true, 3, NU0, 64, 16>{acc1, bq, PA[0], PA[1], &sv0, 4});
    f16_pass<2, 1, 0, 0, 0, 0, true, 0>(ws, Ppe[0], Ppe[1], acc1, bq, baddr, EpiFwd<SAVE, 2, 4, 0, false, true, 3, NU0, 64, 16>{acc0, bq, PA[0], PA[1], &sv0, 8});

    DMN_F16_STAMP(2);
    // ---- mlps.1 .. mlps.7: layer X reads A and writes B, layer Y reads B and writes A; pass p accumulates out-blocks 2p, 2p+1
    // in set p & 1 and reads its own bias quads, while the other set (pass p - 1, or the previous layer's pass 3) is
    // post-processed into its plane words (and, SAVE, stored as h_l).  LAYER = index of the layer that is being accumulated.
    auto layer = [&](auto lc, unsigned (&Pin)[2][64], unsigned (&Pout)[2][64], auto hasq_prev) __attribute__((always_inline)) {
        constexpr int LAYER = decltype(lc)::value;
        constexpr bool PE_ON = LAYER == 5;                               // skip concat [h, pts] (dm_nerf.py:87)
        constexpr bool HQP = decltype(hasq_prev)::value;                 // (mlps.0 has no table bias)
        const SaveCtx svp = save_ctx(SL.h + (int64_t)(LAYER - 1) * 256 * MP, 256, 4, 0);      // the previous layer's outputs
        const SaveCtx svl = save_ctx(SL.h + (int64_t)LAYER * 256 * MP, 256, 4, 0);
        f16_pass<2, 4, 0, 8, 0, 0, true, VMT>(ws, Pin[0], Pin[1], acc0, bq, baddr,
                                              EpiFwd<SAVE, 2, 6, 8, HQP, true, 3, 1, 64, 16>{acc1, bq, Pin[0], Pin[1], &svp, (LAYER - 1) * 1024 + 12});
        baddr += 256;
        if constexpr (PE_ON) f16_pass<2, 1, 0, 0, 0, 96, false, VMT>(ws, Ppe[0], Ppe[1], acc0, bq, baddr, NoSideC{});
        f16_pass<2, 4, 0, 8, 8, 0, true, VMT>(ws, Pin[0], Pin[1], acc1, bq, baddr,
                                              EpiFwd<SAVE, 2, 0, 0, true, true, 3, 1, 64, 16>{acc0, bq, Pout[0], Pout[1], &svl, LAYER * 1024 + 0});
        baddr += 256;
        if constexpr (PE_ON) f16_pass<2, 1, 0, 0, 0, 96, false, VMT>(ws, Ppe[0], Ppe[1], acc1, bq, baddr, NoSideC{});
        f16_pass<2, 4, 0, 8, 0, 0, true, VMT>(ws, Pin[0], Pin[1], acc0, bq, baddr,
                                              EpiFwd<SAVE, 2, 2, 8, true, true, 3, 1, 64, 16>{acc1, bq, Pout[0], Pout[1], &svl, LAYER * 1024 + 4});
        baddr += 256;
        if constexpr (PE_ON) f16_pass<2, 1, 0, 0, 0, 96, false, VMT>(ws, Ppe[0], Ppe[1], acc0, bq, baddr, NoSideC{});
        f16_pass<2, 4, 0, 8, 8, 0, true, VMT>(ws, Pin[0], Pin[1], acc1, bq, baddr,
                                              EpiFwd<SAVE, 2, 4, 0, true, true, 3, 1, 64, 16>{acc0, bq, Pout[0], Pout[1], &svl, LAYER * 1024 + 8});
        baddr += 256;
        if constexpr (PE_ON) f16_pass<2, 1, 0, 0, 0, 96, false, VMT>(ws, Ppe[0], Ppe[1], acc1, bq, baddr, NoSideC{});
    };
    // straight-line: the group index, hence the ring slot and the hand-over parity, are compile-time constants and the whole
    // network is one basic block -- no control-flow merge at which the register allocator could copy a tile in flight
    typedef std::true_type T_;
    layer(std::integral_constant<int, 1>{}, PA, PB, std::false_type{});
    layer(std::integral_constant<int, 2>{}, PB, PA, T_{});
    layer(std::integral_constant<int, 3>{}, PA, PB, T_{});
    layer(std::integral_constant<int, 4>{}, PB, PA, T_{});
    layer(std::integral_constant<int, 5>{}, PA, PB, T_{});
    layer(std::integral_constant<int, 6>{}, PB, PA, T_{});
    layer(std::integral_constant<int, 7>{}, PA, PB, T_{});

    DMN_F16_STAMP(3);
    // ---- heads on h_7 = planes B (its last two out-blocks arrive under the first groups of the rgb hidden layer)
    f32x16 accR[4], accI[4], accO[1], accD[1], accL[OBX];
    unsigned G1[2][32], G2[2][32];
    const SaveCtx sv7 = save_ctx(SL.h + (int64_t)7 * 256 * MP, 256, 4, 0);
    const SaveCtx svg1 = save_ctx(SL.g1, 128, 2, 2048), svg2 = save_ctx(SL.g2, 128, 2, 2176);
    // rgb hidden' = relu(W' h + W_dirs dirs + b')   (rgb_feature_linear folded in)
    f16_pass<4, 8, 0, 0, 0, 0, true, 0>(ws, PB[0], PB[1], accR, bq, baddr,
                                        EpiFwd<SAVE, 2, 6, 8, true, true, 3, 1, 64, 16>{acc1, bq, PB[0], PB[1], &sv7, 7 * 1024 + 12});
    f16_pass<4, 1, 0, 0, 0, 192, false, 0>(ws, Pde[0], Pde[1], accR, bq, baddr, NoSideC{});
    // ins hidden' = relu(W'' h + b''); reads the rgb hidden layer's bias quads and carries its epilogue (g1)
    f16_pass<4, 8, 0, 16, 0, 0, true, 0>(ws, PB[0], PB[1], accI, bq, baddr,
                                         EpiFwd<SAVE, 4, 0, 0, true, true, 24, 1, 32, 16>{accR, bq, G1[0], G1[1], &svg1, 0});
    baddr += 512;
    // rgb_linear (dm_nerf.py:102) on the rgb hidden planes, then density_linear (:101) on h_7: together they carry the ins
    // hidden layer's epilogue (g2), whose bias quads the first of them reads
    f16_pass<1, 1, 0, 16, 0, 0, true, 0>(ws, G1[0], G1[1], accO, bq, baddr, NoSideC{});
    f16_pass<1, 2, 0, 0, 0, 24, true, 0>(ws, PB[0], PB[1], accD, bq, baddr,
                                         EpiFwd<SAVE, 4, 0, 0, true, true, 24, (SAVE ? 3 : 2), 32, 16>{accI, bq, G2[0], G2[1], &svg2, 0});
    // ins_linear (:103)
    f16_pass<OBX, OBX, 0, 0, 0, 0, true, 0>(ws, G2[0], G2[1], accL, bq, baddr, NoSideC{});

    DMN_F16_STAMP(4);
    // ---- outputs: cat[rgb, density, ins] (dm_nerf.py:105); biases of the three output layers from the table
    float* __restrict__ out_row = a.raw + m * (4 + C);
    if (valid) {
        const float* bt = tab + half * 16;
        if (half == 0) {
            out_row[0] = accO[0][0] + bt[F16_TAB_RGBO + 0];
            out_row[1] = accO[0][1] + bt[F16_TAB_RGBO + 1];
            out_row[2] = accO[0][2] + bt[F16_TAB_RGBO + 2];
            out_row[3] = accD[0][0] + bt[F16_TAB_DEN];
        }
#pragma unroll
        for (int b = 0; b < OBX; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (ch < C) out_row[4 + ch] = accL[b][r] + bt[F16_TAB_INSO + b * 32 + r];
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the last (landing-zone) fetches
    DMN_F16_STAMP(5);
}

}  // namespace
