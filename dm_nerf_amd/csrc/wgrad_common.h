// wgrad_common.h -- shared by wgrad.hip (f32 MFMA) and wgrad_split.hip (opt-in split-bf16 MFMA): the job / output tables of
// the split-K plan, the wave partition of an output tile, the LDS ring geometry and the partial-tile epilogue.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "layout.h"
#include "mlp_common.h"

using namespace dmn;

// wgrad.hip: second stage (fixed-order sum of the per-slice partials) + the feature-linear gradients from G / Q (heads.hip)
int dmn_wgrad_finish(const void* d_outs, int n_outs, const float* d_params_flat, int ins_num, float* d_part, float* d_grad_flat, hipStream_t stream,
                     const float* d_unscale = nullptr);

namespace {

constexpr int KT = 32;                      // samples per chunk
constexpr int WG_LDS_BYTES = 147456;        // ring budget (of the CU's 160 KiB)
constexpr int MAX_DEPTH = 6;                // ring slots (the chunk loop is unrolled by the depth)

// One work item = one (job, sample slice) with its own partial tile (device table, offsets only => reusable across steps).
// A WORKGROUP runs one item or a few: the plan fills every workgroup to the same TIME, so the workgroup that finishes a job's
// last slice early continues with the first slice of the next job.  The table holds one LEADER item per workgroup first (the
// grid: one block per leader -- never more blocks than CUs: blocks go to the XCDs round-robin, and a 257th would wait a whole
// launch behind the 32 resident ones of its XCD), then the follower items; a leader names its `follow` followers from `next` on.
struct WgJob {
    int64_t a_off, b_off;     // float offsets of the A / B TENSORS inside their source buffers
    int a_R, b_R;             // total rows of those tensors (block stride = R*32 floats)
    int a_row0, b_row0;       // first row of the job inside the tensor
    int64_t part_off;         // float offset of this workgroup's partial [NBA*32][NBB*32] in the workspace
    int64_t bias_off;         // float offset of its partial row sums [NBA*32], or -1
    int a_src, b_src;         // 0 = saved activations, 1 = dgrad output, 2 = transposed d raw
    int rowsA, rowsB;         // valid rows (the rest of the 32-row blocks is zero)
    int cls;                  // shape class (NBA, NBB)
    int chunk0, nchunk;       // 32-sample chunks [chunk0, chunk0 + nchunk)
    int follow;               // leader: number of further items the same workgroup runs; follower: -1
    int next;                 // leader with followers: index of the first of them (they are consecutive); else -1
    int pad;
};
static_assert(sizeof(WgJob) % 8 == 0, "WgJob layout");

// One output tensor slice (weight columns [col_off, col_off + rowsB) of a parameter, plus its bias).
struct WgOut {
    int64_t part_off, slice_stride;   // first partial, distance between slices
    int64_t bias_part_off, bias_slice_stride;   // per slice: bias_sub partial vectors of NBA*32 floats each
    int64_t out_off, bias_out_off;    // float offsets into the flat gradient vector (reference order); bias -1 = none
    int n_slices, rowsA, rowsB, ldp;  // ldp = NBB*32
    int ld_out, col_off, bias_sub, ldb;        // bias_sub shares per slice, ldb = NBA*32 apart
    int perm_a, perm_b, to_scratch, bias_split;   // operand rows are in the accumulator-layout memory order (layout.h::row_feature);
                                               // to_scratch: out_off addresses the head of the partials workspace (G / Q of heads.hip)
    int64_t bias_out_off2;                     // bias_split > 0: the row sums of rows >= bias_split go here (the [dg1 ; dg2] job: two layers' biases)
};

struct WgArgs {
    const float* src[3];
    float* part;
    const WgJob* jobs;
    int64_t Mp;
    long long* trace;     // diagnostic: per-workgroup {start, end} of the 100 MHz wall clock, or null
};

template <int NBA, int NBB>
struct Split {   // which (A block, B block) pairs a wave owns: a rectangle SAn x SBn; block = literal + wave part
    static constexpr int SBn = NBB >= 4 ? NBB / 4 : 1;
    static constexpr int SAn = NBB >= 4 ? NBA : (NBB == 2 ? NBA / 2 : NBA / 4);
    static constexpr int NSHARE = NBB >= 4 ? 4 : (NBB == 2 ? 2 : 1);        // waves that hold the same A blocks
    static constexpr int HOST_NSHARE = NSHARE;                              // row-sum shares the plan reserves per slice (wgrad.hip::make_plan)
    static_assert(NBB >= 4 || (NBB == 2 && NBA % 2 == 0) || (NBB == 1 && NBA % 4 == 0), "unsupported shape class");
    // A block of (wave w, k) = a_lit(k) + a_wave(w); B block = b_lit(k) + b_wave(w)
    __device__ static constexpr int a_lit(int k) { return NBB >= 4 ? k : (NBB == 2 ? 2 * k : 4 * k); }
    __device__ static int a_wave(int w) { return NBB >= 4 ? 0 : (NBB == 2 ? (w >> 1) : w); }
    __device__ static constexpr int b_lit(int k) { return NBB >= 4 ? 4 * k : 0; }
    __device__ static int b_wave(int w) { return NBB >= 4 ? w : (NBB == 2 ? (w & 1) : 0); }
    __device__ static int share_rank(int w) { return NBB >= 4 ? w : (NBB == 2 ? (w & 1) : 0); }
};

template <int NBA, int NBB>
struct Ring {
    static constexpr int BUF = (NBA + NBB) * 4096;                                    // bytes per chunk: rows * 128
    static constexpr int D = WG_LDS_BYTES / BUF < MAX_DEPTH ? WG_LDS_BYTES / BUF : MAX_DEPTH;
    static_assert(D >= 2, "ring needs two slots");
};

// Epilogue of one workgroup: the partial tile [NBA*32][NBB*32] (C layout: a lane holds column j = li, rows crow(r, half))
// and this wave's share of the row sums of A.
template <class SP, int NBA, int NBB, int NPAIR, int SAn>
__device__ __forceinline__ void store_partials(const WgArgs& a, const WgJob& jb, const f32x16 (&acc)[NPAIR], const f32x4 (&bs)[SAn],
                                               int w, int half, int li, bool want_bias, int my_rank) {
    constexpr int SBn = SP::SBn;
    static_assert(SAn == SP::SAn && NPAIR == SAn * SBn, "tile partition");
    float* __restrict__ P = a.part + jb.part_off;
    constexpr int LDP = NBB * 32;
#pragma unroll
    for (int ia = 0; ia < SAn; ++ia) {
        const int ba = SP::a_lit(ia) + SP::a_wave(w);
#pragma unroll
        for (int ib = 0; ib < SBn; ++ib) {
            const int bb = SP::b_lit(ib) + SP::b_wave(w);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = ba * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                P[(int64_t)row * LDP + bb * 32 + li] = acc[ia * SBn + ib][r];
            }
        }
        if (want_bias) {
            // this wave's share of the row sums (its rounds, both k halves); the reduction adds the NSHARE shares
            float sum = (bs[ia][0] + bs[ia][1]) + (bs[ia][2] + bs[ia][3]);
            sum += __shfl_xor(sum, 32);
            if (half == 0) {
                a.part[jb.bias_off + (int64_t)my_rank * (NBA * 32) + ba * 32 + li] = sum;
#pragma unroll
                for (int sh = SP::NSHARE; sh < SP::HOST_NSHARE; sh += SP::NSHARE)       // shares of the plan this partition does not use
                    a.part[jb.bias_off + (int64_t)(my_rank + sh) * (NBA * 32) + ba * 32 + li] = 0.f;
            }
        }
    }
}


// shape classes (NBA, NBB)
enum { C_8_8 = 0, C_4_8, C_8_2, C_4_1, C_1_8, C_1_4, C_2_4, C_3_4, C_4_4, N_CLASSES };
constexpr int CLS_NBA[N_CLASSES] = {8, 4, 8, 4, 1, 1, 2, 3, 4};
constexpr int CLS_NBB[N_CLASSES] = {8, 8, 2, 1, 8, 4, 4, 4, 4};

}  // namespace
