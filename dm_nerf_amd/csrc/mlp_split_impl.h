// mlp_split_impl.h -- OPT-IN kernel (entry points: mlp_split.hip inference, mlp_split_train.hip training forward): the DM-NeRF MLP on v_mfma_f32_32x32x16_bf16 with every f32 operand
// split into three bf16 planes ("bf16x3"), six products per term pair accumulated in f32.
//
//   x = hi + mid + lo  (truncation split: exact for any f32),   w likewise (pre-split weights, layout.h::SplitLayout)
//   w x  ~=  w_hi x_hi + w_hi x_mid + w_mid x_hi + w_hi x_lo + w_mid x_mid + w_lo x_hi      (dropped terms < 2^-24 |w x|)
//
// Every product of two bf16 is exact in f32 and the MFMA accumulates in f32, so the result is in the rounding class
// of an f32 GEMM (emulated through the whole network: max relative error vs float64 1.4e-7, f32 sgemm 2.5e-7), while
// the bf16 MFMA runs 16x the f32 MFMA rate: six terms = 2.7x fewer MFMA cycles.  It is NOT the bitwise fmaf chain of
// the default kernel, hence opt-in (args.mfma_split), like the head fusion it builds on (weights.py::fuse_heads:
// rgb_feature_linear / ins_feature_linear folded into the hidden layers, which keeps a single plane set live).
//
// Structure = the default kernel's (mlp_fwd_impl.h): one wave owns 32 samples for the whole network; a lane's 8
// accumulator registers r = 8 t + q of out-block b ARE its 8 k-slots of k-block 2 b + t of the next layer, so the
// activations never leave registers -- they are kept as three planes of packed bf16 pairs (192 VGPR for 256
// features).  Weights stream through a 3-slot LDS ring of 48 KiB slots (two k-blocks x three planes x eight
// out-blocks) by MUBUF LDS-DMA two slots ahead; A tiles are ds_read_b128, one group = (k-block, plane) = OB tiles
// followed by 3 / 2 / 1 MFMAs per tile (plane hi / mid / lo), reads one group ahead, hand-over at the start of a
// slot's last group.  LDS: 3 x 48 KiB + the 16 KiB table = the CU's full 160 KiB.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "layout.h"
#include "mlp_common.h"
#include "split_bf16.h"

using namespace dmn;

namespace {

constexpr int SP_SLOT_BYTES = SPLIT_SLOT_WORDS * 4;          // 49 152
constexpr int SP_RING_FLOATS = 3 * SPLIT_SLOT_WORDS;
constexpr int SP_LDS_FLOATS = SP_RING_FLOATS + TAB_FLOATS;   // 163 840 bytes
constexpr int SP_DMA = 12;                                   // LDS-DMA pieces (1 KiB) per wave per slot

struct SplitArgs {
    const float* blob;      // [table f32 | split stream]
    BlobLayout L;           // table offsets (the fused f32 layout's)
    SplitLayout S;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    float* raw;
    float* save;            // SAVE: the training workspace of layout.h::SaveLayout (what mlp_bwd.hip / wgrad.hip consume)
    int64_t M;
    int Sr;                 // samples per ray
};

struct SStream {
    rsrc_t rs;
    unsigned voff;          // lane*16 + wave*1024
    float* ring;
    int wave;
    unsigned off;           // byte offset (from the blob start) of the next slot to FETCH
    int cslot;              // ring slot (0..2) of the stream slot being consumed
    f32x4 pre[8];           // first group's A tiles of the next stream slot
};

__device__ __forceinline__ void ss_fetch_piece(const SStream& ws, int target_slot, int i) {
    float* dst = ws.ring + target_slot * SPLIT_SLOT_WORDS + ws.wave * 256 + i * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ws.rs, (DMN_LAS void*)dst, 16, (int)ws.voff, (int)(ws.off + i * 4096), 0, 0);
}

// planes of NV accumulator-layout f32x16 blocks: block b, register r = 8 t + q -> k-block 2 b + t, word (q >> 1)
template <int NV, bool RELU>
__device__ __forceinline__ void split_blocks(const f32x16 (&x)[NV], unsigned (&P)[3][NV * 8]) {
#pragma unroll
    for (int b = 0; b < NV; ++b)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float x0 = RELU ? relu1(x[b][r]) : x[b][r], x1 = RELU ? relu1(x[b][r + 1]) : x[b][r + 1];
            const int w = (2 * b + (r >> 3)) * 4 + ((r & 7) >> 1);
            split_pair(x0, x1, P[0][w], P[1][w], P[2][w]);
        }
}

// One GEMM segment: NKB k-blocks of B planes P (words 4 kb .. 4 kb + 3 of each plane, kb = KB0 ..), OB out-blocks.
// NEXT_OB: out-blocks of the segment that follows in the stream (0 = none).
// Per k-block three groups hi, mid, lo of OB tiles each, with 3 / 2 / 1 MFMAs per tile.  Reads (inline asm, one per MFMA
// gap) run ahead of their use so that no group waits for LDS: the hi group issues the reads of the mid AND lo tiles of
// its k-block (2 OB reads in 3 OB gaps), the mid group those of the next k-block's hi tiles (into ws.pre, which is the
// hi buffer), the short lo group none.  The slot hand-over therefore sits at the start of the LAST k-block's mid group.
template <int KB0, int NKB, int OB, int NEXT_OB, int NW>
__device__ __forceinline__ void gemm_split(SStream& ws, const unsigned (&P)[3][NW], f32x16 (&acc)[OB], int lane) {
    constexpr int KPS = split_kb_per_slot(OB);
    constexpr int NSLOT = split_slots(NKB, OB);
    static_assert(NW >= (KB0 + NKB) * 4, "B planes too small");
    static_for<NSLOT>([&](auto sc) {
        constexpr int sl = decltype(sc)::value;
        constexpr int KBN = (NKB - sl * KPS) < KPS ? (NKB - sl * KPS) : KPS;        // k-blocks in this slot
        constexpr bool LAST_SLOT = sl == NSLOT - 1;
        constexpr int NXT = LAST_SLOT ? NEXT_OB : OB;                               // hi tiles of the next slot's first k-block
        constexpr int GAPS_BEFORE_HANDOVER = (KBN * 6 - 3) * OB;                    // MFMA gaps before the last k-block's mid group
        constexpr int PD = (GAPS_BEFORE_HANDOVER - OB) / SP_DMA >= 1 ? (GAPS_BEFORE_HANDOVER - OB) / SP_DMA : 1;
        const unsigned s0 = lds_addr(ws.ring + ws.cslot * SPLIT_SLOT_WORDS) + lane * 16;
        const int nslot = ws.cslot == 2 ? 0 : ws.cslot + 1;
        const int fslot = ws.cslot == 0 ? 2 : ws.cslot - 1;                          // ring slot released by the last hand-over: target of slot + 2
        const unsigned s1 = lds_addr(ws.ring + nslot * SPLIT_SLOT_WORDS) + lane * 16;
        f32x4 am[OB], al[OB];                                                        // mid / lo tiles (hi tiles live in ws.pre)
        static_for<KBN>([&](auto kc) {
            constexpr int kbl = decltype(kc)::value;
            constexpr int kb = KB0 + sl * KPS + kbl;
            constexpr bool LASTK = kbl == KBN - 1;
            constexpr int G0 = kbl * 6 * OB;                                         // gaps of this slot before this k-block
            auto dma_at = [&](auto ggc) {                                            // the SP_DMA pieces of stream slot + 2
                constexpr int GG = decltype(ggc)::value;
                static_for<SP_DMA>([&](auto pc) {
                    constexpr int k = decltype(pc)::value;
                    constexpr int at = OB + k * PD < GAPS_BEFORE_HANDOVER ? OB + k * PD : GAPS_BEFORE_HANDOVER - 1;
                    if constexpr (at == GG) ss_fetch_piece(ws, fslot, k);
                });
            };
            // ---- hi group: 3 OB MFMAs; reads: this k-block's mid tiles, then its lo tiles
            lds_wait<0>(ws.pre);
            __builtin_amdgcn_sched_barrier(0);
            static_for<3 * OB>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int t = i / OB, ob = i % OB;
                if constexpr (i < OB) lds_read16_async<((kbl * 3 + 1) * OB + i) * 1024>(am[i], s0);
                else if constexpr (i < 2 * OB) lds_read16_async<((kbl * 3 + 2) * OB + (i - OB)) * 1024>(al[i - OB], s0);
                dma_at(std::integral_constant<int, G0 + i>{});
                acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_a(ws.pre[ob]), as_b(&P[t][kb * 4]), acc[ob], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            // ---- mid group: 2 OB MFMAs; reads: the next k-block's hi tiles (next slot's after the hand-over)
            if constexpr (LASTK && NXT > 0) {
                lds_wait<0>(am);                       // every read of this slot has returned before the slot is released
#pragma unroll
                for (int i = 0; i < OB; ++i) asm volatile("" : "+" DMN_TILE_RC(al[i]));
            } else {
                lds_wait<OB>(am);                      // the OB lo reads issued after the mid reads may still be in flight
            }
            if constexpr (LASTK && NXT > 0) {
                // hand-over: the next stream slot has landed (vmcnt retires in order: the SP_DMA younger pieces belong to
                // the slot after it) in every wave's view, and every wave has issued all its reads of this slot but the
                // lo tiles of this k-block, which were issued one group ago and are waited for below
                __builtin_amdgcn_s_waitcnt(0x0F70 | (SP_DMA & 15) | ((SP_DMA >> 4) << 14));
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<2 * OB>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int t = i / OB, ob = i % OB;
                if constexpr (!LASTK && i < OB) lds_read16_async<(((kbl + 1) * 3) * OB + i) * 1024>(ws.pre[i], s0);
                if constexpr (LASTK && i < NXT) lds_read16_async<i * 1024>(ws.pre[i], s1);
                if constexpr (!LASTK) dma_at(std::integral_constant<int, G0 + 3 * OB + i>{});
                acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_a(am[ob]), as_b(&P[t][kb * 4]), acc[ob], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            // ---- lo group: OB MFMAs, no reads
            lds_wait<(LASTK ? (NXT < OB ? NXT : OB) : OB)>(al);    // the next hi tiles (issued after the lo reads) may still be in flight
            __builtin_amdgcn_sched_barrier(0);
            static_for<OB>([&](auto ic) {
                constexpr int ob = decltype(ic)::value;
                if constexpr (!LASTK) dma_at(std::integral_constant<int, G0 + 5 * OB + ob>{});
                acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_a(al[ob]), as_b(&P[0][kb * 4]), acc[ob], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        // the carry tiles were read through inline asm: make them real values before any other code (or the register
        // allocator) may touch them
        if constexpr (NXT > 0) lds_wait<0>(ws.pre);
        ws.off += SP_SLOT_BYTES;
        ws.cslot = nslot;
    });
}

// SAVE (opt-in training forward): one accumulator block of relu outputs -> its 16 rows of a block-major training tensor
// (TID-addressed stores, mlp_common.h::RowIO) and its 16 mask bits (word b >> 1, same bit order as pack_mask)
__device__ __forceinline__ void save_block(const RowIO& io, int b, const f32x16& v, unsigned& mword) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float x = relu1(v[r]);
        DMN_ACT_STORE_B32(f2u(x), io.rs, run_off(0, r), (int)(io.soff + b * 4096), DMN_STORE_AUX);
        asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mword) : "v"(x) : "vcc");
    }
}

template <int OBX, bool SAVE = false>
__global__ __launch_bounds__(256) void mlp_split_kernel(const SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [ring 3 x 48 KiB][table 16 KiB]
    float* const tab = lds + SP_RING_FLOATS;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nblk = (a.M + 31) / 32;
    const int64_t blk_raw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t blk = blk_raw < nblk ? blk_raw : nblk - 1;             // (a wave beyond the batch duplicates the last block)
    const int64_t m_raw = blk * 32 + (lane & 31);
    const bool valid = m_raw < a.M;
    const int64_t m = valid ? m_raw : a.M - 1;
    const BlobLayout& L = a.L;

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
    SStream ws;
    ws.rs = uniform_rsrc(a.blob, a.S.total);
    ws.wave = wave;
    ws.voff = (unsigned)(lane * 16 + wave * 1024);
    ws.ring = lds;
    ws.off = __builtin_amdgcn_readfirstlane((unsigned)(a.S.stream * 4));
    ws.cslot = 0;
    // prologue: stream slots 0 and 1 into ring slots 0 and 1
#pragma unroll
    for (int i = 0; i < SP_DMA; ++i) ss_fetch_piece(ws, 0, i);
    ws.off += SP_SLOT_BYTES;
#pragma unroll
    for (int i = 0; i < SP_DMA; ++i) ss_fetch_piece(ws, 1, i);
    ws.off += SP_SLOT_BYTES;                                             // from now on `off` = consumed slot + 2

    f32x16 pe[2], de[1];
    encode<POS_L, 2>(pt, pe, half);
    encode<DIR_L, 1>(vd, de, half);

    // slot 0 landed (the SP_DMA pieces of slot 1 may still fly), table visible; first tiles of slot 0
    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const unsigned s0 = lds_addr(ws.ring) + lane * 16;
        static_for<8>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_read16_async<i * 1024>(ws.pre[i], s0); });
        lds_wait<0>(ws.pre);
    }

    f32x16 acc[8];
    unsigned Ph[3][64];                        // the 256 trunk features as three planes of bf16 pairs
    float sigma = 0.f, rgb_out[3] = {0.f, 0.f, 0.f};
    float* __restrict__ out_row = a.raw + m * (4 + L.C);
    // SAVE: the saved tensors are f32 (the backward kernels are the f32 ones); h_l is written where it is split into planes
    const SaveLayout SL = make_save_layout(a.M);
    const int64_t MP = save_row_len(a.M);
    rsrc_t bits_rs;
    if constexpr (SAVE) bits_rs = uniform_rsrc(a.save + SL.bits, (int64_t)(BITS_WORDS_PER_BLOCK / 32) * MP);
    auto save_h = [&](int layer) {
        const RowIO hio = make_rowio(a.save + SL.h + (int64_t)layer * 256 * MP, 256, MP, blk, lane);
        unsigned mw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int b = 0; b < 8; ++b) save_block(hio, b, acc[b], mw[b >> 1]);
        const u32x4 v = {mw[0], mw[1], mw[2], mw[3]};
        __builtin_amdgcn_raw_buffer_store_b128(v, bits_rs, (int)((blk * BITS_WORDS_PER_BLOCK + lane * 4) * 4) + layer * 1024, 0, 0);
    };
    auto save_hidden = [&](const f32x16 (&hid)[4], int64_t tensor_off, int bit_word0) {      // g1 / g2: 128 rows, 2 mask words
        const RowIO gio = make_rowio(a.save + tensor_off, 128, MP, blk, lane);
        unsigned mw[2] = {0u, 0u};
#pragma unroll
        for (int b = 0; b < 4; ++b) save_block(gio, b, hid[b], mw[b >> 1]);
        __builtin_amdgcn_raw_buffer_store_b32(mw[0], bits_rs, (int)((blk * BITS_WORDS_PER_BLOCK + bit_word0 + lane * 2) * 4), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(mw[1], bits_rs, (int)((blk * BITS_WORDS_PER_BLOCK + bit_word0 + lane * 2 + 1) * 4), 0, 0);
    };
    if constexpr (SAVE) {
        store_encoded_rows<POS_L, 2>(a.save + SL.pe, MP, blk, lane, pe);
        store_encoded_rows<DIR_L, 1>(a.save + SL.de, MP, blk, lane, de);
    }

    // ---- mlps.0 : 63 -> 256
    {
        unsigned Pp[3][16];
        split_blocks<2, false>(pe, Pp);
        init_bias_lds<8>(tab + L.b0, acc, half);
        gemm_split<0, 4, 8, 8>(ws, Pp, acc, lane);
    }
    if constexpr (SAVE) save_h(0);
    split_blocks<8, true>(acc, Ph);

    // ---- mlps.1 .. mlps.7 (skip concat [h, pts] into mlps.5, dm_nerf.py:87)
#pragma nounroll
    for (int st = 0; st < 7; ++st) {
        init_bias_lds<8>(tab + L.b_stage + st * (int)bias_floats(8), acc, half);
        gemm_split<0, 16, 8, 8>(ws, Ph, acc, lane);
        if (st == 4) {
            unsigned Pp[3][16];
            split_blocks<2, false>(pe, Pp);
            gemm_split<0, 4, 8, 8>(ws, Pp, acc, lane);
        }
        if (st == 6) {
            // density_linear (dm_nerf.py:101) from the f32 activations, before they are split
            const f32x4* wd = reinterpret_cast<const f32x4*>(tab + L.w_den + half * 128);
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const f32x4 w = wd[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = 4 * i + j;
                    part = fmaf(relu1(acc[p >> 4][p & 15]), w[j], part);
                }
            }
            sigma = part + __shfl_xor(part, 32) + tab[L.b_den];
        }
        if constexpr (SAVE) save_h(st + 1);
        split_blocks<8, true>(acc, Ph);
    }

    // ---- rgb branch: hidden = relu(W' h + W_dirs dirs + b')   (rgb_feature_linear folded in)
    {
        f32x16 hid[4];
        init_bias_lds<4>(tab + L.b_rgbh, hid, half);
        gemm_split<0, 16, 4, 4>(ws, Ph, hid, lane);
        {
            unsigned Pd[3][8];
            split_blocks<1, false>(de, Pd);
            gemm_split<0, 2, 4, 4>(ws, Pd, hid, lane);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) hid[b] = relu16(hid[b]);
        if constexpr (SAVE) save_hidden(hid, SL.g1, 2048);
#pragma unroll
        for (int c = 0; c < 3; ++c) {                                   // rgb_linear (dm_nerf.py:102) on the VALU
            const f32x4* wr = reinterpret_cast<const f32x4*>(tab + L.w_rgbo + (c * 2 + half) * 64);
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
        }
    }
    // ---- ins branch: hidden = relu(W'' h + b''), logits = ins_linear(hidden)
    {
        f32x16 hid[4];
        init_bias_lds<4>(tab + L.b_insh, hid, half);
        gemm_split<0, 16, 4, OBX>(ws, Ph, hid, lane);
        if constexpr (SAVE) save_hidden(hid, SL.g2, 2176);
        unsigned Pi[3][32];
        split_blocks<4, true>(hid, Pi);
        f32x16 io[OBX];
        init_bias_lds<OBX>(tab + L.b_inso, io, half);                     // (OBX > OBI: the extra block's bias slots are zero)
        gemm_split<0, 8, OBX, 0>(ws, Pi, io, lane);
        if (valid) {
#pragma unroll
            for (int b = 0; b < OBX; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (ch < L.C) out_row[4 + ch] = io[b][r];
                }
        }
    }
    if (valid && half == 0) {
        out_row[0] = rgb_out[0]; out_row[1] = rgb_out[1]; out_row[2] = rgb_out[2]; out_row[3] = sigma;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the last (landing-zone) fetches
}

}  // namespace
