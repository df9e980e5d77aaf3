// wgrad.hip -- weight / bias gradients of the DM-NeRF MLP for gfx950:  dW = dy . x^T  over the batch.
//
// Both operands are the block-major [32-sample block][rows][32] tensors the training forward
// (x: layer inputs) and the dgrad pass (dy: pre-activation gradients) wrote, so every job is an "NT"
// GEMM over K = samples (up to ~10^6) whose output is at most 256 x 256, and the operands of one
// 32-sample chunk are two CONTIGUOUS tiles of rows*128 bytes:
//   * split-K: the samples are cut into slices, one workgroup per (job, slice); slice counts are
//     proportional to the job's measured chunk time so that one wave of workgroups (<= 256 CUs, one per
//     CU) is balanced; per-slice partials go to a workspace and a second kernel adds them in fixed order
//     (deterministic, no float atomics);
//   * a chunk's two tiles go global -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction)
//     into a D-deep ring (D = 2 for the 256 x 256 jobs, up to 6 for the skinny ones, whose chunks are
//     shorter than the HBM latency), rows of 128 bytes with their 16-byte units XOR-swizzled by
//     (row >> 1) & 7 so that the ds_read_b128 of 32 different rows is bank-conflict free;
//   * the 4 waves run v_mfma_f32_32x32x2_f32 with A[i][k] = dy[row i][sample k], B[k][j] = x[row j][sample k];
//     a lane reads 4 consecutive samples of its row per ds_read_b128 (k = 0 / 1 <-> lane halves);
//   * the inner loop is hand-scheduled like the MLP kernels (mlp_common.h): operand reads one round
//     ahead through inline asm, one per MFMA gap; the ring hand-over (vmcnt + barrier) at the start of a
//     chunk's last round; no vector-ALU work besides the bias sums (every VALU instruction of a
//     one-wave-per-SIMD kernel is paid for in MFMA time): row sums of dy as 2 v_pk_add_f32 per A block
//     and round, the rounds dealt out over the waves that hold the same A blocks.
// Exact f32 (fmaf-chain MFMA), no vendor BLAS.  Roofline: ~equal parts MFMA (2 x MAC x M FLOP) and
// HBM (each dy / x row is read once per job: ~23 KB per sample and model).
#include <cmath>
#include <cstring>
#include <vector>

#include "params.h"
#include "wgrad_common.h"

int dmn_head_unfuse(const float* d_flat, int ins_num, const float* d_G, const float* d_Q, float* d_grad, hipStream_t stream);   // heads.hip

namespace {

template <int NBA, int NBB>
__device__ __forceinline__ void run_job(const WgArgs& a, const WgJob& jb, float* lds) {
    typedef Split<NBA, NBB> SP;
    typedef Ring<NBA, NBB> RG;
    constexpr int SAn = SP::SAn, SBn = SP::SBn, NPAIR = SAn * SBn;
    constexpr int NL = NBA + NBB;                  // DMA pieces per wave per chunk (1 KiB each)
    constexpr int NR = SAn + SBn;                  // operand reads per round
    constexpr int NGAP = 4 * NPAIR;                // MFMAs per round
    constexpr int D = RG::D, BUF = RG::BUF;
    static_assert((D - 1) * NL <= 63, "vmcnt range");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));               // opaque per item: the lane geometry below is recomputed for each item of a workgroup
                                               // (hoisted out of the item loop, nine shape classes' worth of it would spill)
    const int lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = lds_addr(lds);

    // ---- DMA geometry: wave w owns the 1-KiB pieces q = w + 4 j of a tile (rows 8 q .. 8 q + 7); lane l lands
    // at LDS row 8 q + (l >> 3), unit l & 7, so it must FETCH unit (l & 7) ^ ((row >> 1) & 7) of that row
    const int drow = 8 * w + (lane >> 3);
    const int dvoff = drow * 128 + (((lane & 7) ^ ((drow >> 1) & 7)) << 4);            // + 4096 j (soffset)
    const float* __restrict__ A = a.src[jb.a_src] + jb.a_off + (int64_t)jb.a_row0 * 32;
    const float* __restrict__ B = a.src[jb.b_src] + jb.b_off + (int64_t)jb.b_row0 * 32;
    const int64_t strideA = (int64_t)jb.a_R * 32, strideB = (int64_t)jb.b_R * 32;     // floats per 32-sample block
    const int nchunk = jb.nchunk;
    // rows beyond rowsA / rowsB are out of range of the tile descriptor: never fetched (what the LDS holds there
    // only reaches outputs the reduction ignores)
    auto dma_chunk_piece = [&](int c, unsigned slot_byte, int i) {      // piece i of NL for chunk c (clamped) into a ring slot
        const int cc = c < nchunk ? c : nchunk - 1;
        const bool isA = i < NBA;
        const int j = isA ? i : i - NBA;
        const float* base = isA ? A + (int64_t)(jb.chunk0 + cc) * strideA : B + (int64_t)(jb.chunk0 + cc) * strideB;
        const rsrc_t rs = uniform_rsrc(base, (int64_t)(isA ? jb.rowsA : jb.rowsB) * 32);
        float* dst = lds + (slot_byte + i * 4096 + w * 1024) / 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (DMN_LAS void*)dst, 16, dvoff, j * 4096, 0, 0);
    };

    // ---- read geometry: lane (li, half) reads row 32 blk + li, unit (2 t + half) ^ ((li >> 1) & 7) in round t
    // (slot 0 addresses; the chunk loop adds the ring slot's byte offset: 10 VALU adds per 256+ MFMAs)
    unsigned offA[4], offB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned o = lds0 + li * 128 + ((((2 * t + half) ^ ((li >> 1) & 7))) << 4);
        offA[t] = o + SP::a_wave(w) * 4096;
        offB[t] = o + (NBA + SP::b_wave(w)) * 4096;
    }

    f32x16 acc[NPAIR];
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) acc[i] = (f32x16)(0.f);
    f32x4 bs[SAn];
#pragma unroll
    for (int i = 0; i < SAn; ++i) bs[i] = (f32x4)(0.f);
    const bool want_bias = jb.bias_off >= 0;
    const int my_rank = SP::share_rank(w);

    f32x4 av[2][SAn], bv[2][SBn];
    auto read_ops_one = [&](auto gc, int buf, unsigned addrA, unsigned addrB) {     // operand g of a round
        constexpr int g = decltype(gc)::value;
        if constexpr (g < SAn) lds_read16_async<SP::a_lit(g) * 4096>(av[buf][g], addrA);
        else lds_read16_async<SP::b_lit(g - SAn) * 4096>(bv[buf][g - SAn], addrB);
    };

    // ---- prologue: D chunks in flight, chunk 0 landed, its round-0 operands on their way
#pragma unroll
    for (int sl = 0; sl < D; ++sl)
#pragma unroll
        for (int i = 0; i < NL; ++i) dma_chunk_piece(sl, sl * BUF, i);
    __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 1) * NL) & 15) | ((((D - 1) * NL) >> 4) << 14));     // vmcnt((D-1) NL) only
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    static_for<NR>([&](auto gc) { read_ops_one(gc, 0, offA[0], offB[0]); });

    unsigned sb = 0;                                    // byte offset of the ring slot of chunk c (uniform)
#pragma nounroll
    for (int c = 0; c < nchunk; ++c) {
        const unsigned nb = sb + BUF == (unsigned)(D * BUF) ? 0u : sb + BUF;
        unsigned cA[4], cB[4];
#pragma unroll
        for (int t = 1; t < 4; ++t) { cA[t] = offA[t] + sb; cB[t] = offB[t] + sb; }
        cA[0] = offA[0] + nb; cB[0] = offB[0] + nb;     // round 0 of the NEXT chunk (read in this chunk's round 3)
        static_for<4>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            lds_wait<0>(av[r & 1]);
#pragma unroll
            for (int k = 0; k < SBn; ++k) asm volatile("" : "+" DMN_TILE_RC(bv[r & 1][k]));
            if constexpr (r == 3) {
                // ring hand-over: chunk c + 1 has landed in every wave's view, and this chunk's slot is released
                __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 2) * NL) & 15) | ((((D - 2) * NL) >> 4) << 14));
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (want_bias && (r % SP::NSHARE) == my_rank) {            // wave-uniform
                asm volatile("");                                       // (keeps this a branch: no if-conversion into selects)
#pragma unroll
                for (int ia = 0; ia < SAn; ++ia) bs[ia] += av[r & 1][ia];
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<NGAP>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int u = g / NPAIR, pr = g % NPAIR, ia = pr / SBn, ib = pr % SBn;
                if constexpr (g < NR) read_ops_one(gc, (r + 1) & 1, cA[(r + 1) & 3], cB[(r + 1) & 3]);
                if constexpr (r == 3) {                                 // refill the released slot with chunk c + D
                    constexpr int G0 = NR < NGAP ? NR : NGAP - 1;
                    constexpr int PD = (NGAP - G0) / NL > 0 ? (NGAP - G0) / NL : 1;
                    static_for<NL>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        constexpr int at = G0 + i * PD < NGAP ? G0 + i * PD : NGAP - 1;
                        if constexpr (at == g) dma_chunk_piece(c + D, sb, i);
                    });
                }
                acc[pr] = mfma32(av[r & 1][ia][u], bv[r & 1][ib][u], acc[pr]);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        sb = nb;
    }
    // the ring's last (clamped) refills and the read-ahead of the chunk after the last one: both land in registers /
    // LDS nobody uses, but they must have landed before the epilogue reuses either (the ties keep the read-ahead's
    // destination registers allocated until then)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < SAn; ++k) asm volatile("" : "+" DMN_TILE_RC(av[0][k]));
#pragma unroll
    for (int k = 0; k < SBn; ++k) asm volatile("" : "+" DMN_TILE_RC(bv[0][k]));

    store_partials<Split<NBA, NBB>, NBA, NBB>(a, jb, acc, bs, w, half, li, want_bias, my_rank);
}

__global__ __launch_bounds__(256) void wgrad_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    WgJob jb = a.jobs[blockIdx.x];
    if (a.trace && threadIdx.x == 0) a.trace[2 * blockIdx.x] = (long long)wall_clock64();
    for (int item = jb.next - 1, more = jb.follow;; --more) {      // (block = leader item; its followers start at jb.next)
        switch (jb.cls) {                              // workgroup-uniform
            case C_8_8: run_job<8, 8>(a, jb, lds); break;
            case C_4_8: run_job<4, 8>(a, jb, lds); break;
            case C_8_2: run_job<8, 2>(a, jb, lds); break;
            case C_4_1: run_job<4, 1>(a, jb, lds); break;
            case C_1_8: run_job<1, 8>(a, jb, lds); break;
            case C_1_4: run_job<1, 4>(a, jb, lds); break;
            case C_2_4: run_job<2, 4>(a, jb, lds); break;
            case C_3_4: run_job<3, 4>(a, jb, lds); break;
            case C_4_4: run_job<4, 4>(a, jb, lds); break;
            default: break;
        }
        if (more <= 0) break;
        __syncthreads();                               // every wave is done with the LDS ring: the next item refills it
        jb = a.jobs[++item];
    }
    if (a.trace && threadIdx.x == 0) a.trace[2 * blockIdx.x + 1] = (long long)wall_clock64();
}

// Adds the per-slice partials in slice order into the flat gradient vector.
// unscale: null, or the factor 2^-s that undoes the power-of-two scaling the split-f16 data-gradient kernel applied to dL/draw
// (every operand of every job on the dy side carries it: the sums are linear in it)
__global__ void wgrad_reduce_kernel(float* __restrict__ part, const WgOut* __restrict__ outs, int n_outs,
                                    float* __restrict__ grad_flat, const float* __restrict__ unscale) {
    const WgOut o = outs[blockIdx.y];
    const float us = unscale ? *unscale : 1.f;
    float* __restrict__ grad = o.to_scratch ? part : grad_flat;        // (the bias of a scratch output still goes to the gradient)
    const int64_t n_w = (int64_t)o.rowsA * o.rowsB;
    const int64_t n_all = n_w + (o.bias_out_off >= 0 ? o.rowsA : 0);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_all; e += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        if (e < n_w) {
            const int r = (int)(e / o.rowsB), c = (int)(e % o.rowsB);
            const float* p = part + o.part_off + (int64_t)r * o.ldp + c;
            // fixed slice order; eight loads in flight per trip (the loop is latency-bound otherwise)
            int k = 0;
            for (; k + 8 <= o.n_slices; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[(k + u) * o.slice_stride];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; k < o.n_slices; ++k) s += p[k * o.slice_stride];
            const int ro = o.perm_a ? row_feature(r) : r, co = o.perm_b ? row_feature(c) : c;
            grad[o.out_off + (int64_t)ro * o.ld_out + o.col_off + co] = s * us;
        } else {
            const int r = (int)(e - n_w);
            const float* p = part + o.bias_part_off + r;
            for (int k = 0; k < o.n_slices; ++k)
                for (int q = 0; q < o.bias_sub; ++q) s += p[k * o.bias_slice_stride + (int64_t)q * o.ldb];
            const int rr = o.perm_a ? row_feature(r) : r;
            if (o.bias_split > 0 && rr >= o.bias_split) grad_flat[o.bias_out_off2 + rr - o.bias_split] = s * us;
            else grad_flat[o.bias_out_off + rr] = s * us;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host: the plan
// ------------------------------------------------------------------------------------------
struct JobDesc {
    int a_src; int64_t a_base; int a_R, a_row0;   // tensor: base (in rows of the SaveLayout), its total rows, first row used
    int b_src; int64_t b_base; int b_R, b_row0;
    int rowsA, rowsB, cls;
    int64_t out_off; int ld_out, col_off;
    int64_t bias_out_off;       // -1: bias handled by another job with the same A
    int bias_split = 0; int64_t bias_out_off2 = -1;   // rows >= bias_split: their row sums belong to another layer's bias
};

int class_for(int rowsA, int rowsB) {
    const int nba = (rowsA + 31) / 32, nbb = (rowsB + 31) / 32;
    for (int c = 0; c < N_CLASSES; ++c)
        if (CLS_NBA[c] == nba && CLS_NBB[c] == nbb) return c;
    return -1;
}

struct Plan {
    std::vector<WgJob> jobs;
    std::vector<WgOut> outs;
    int64_t part_floats = 0;
    int n_wgs = 0;              // leaders = blocks of the launch; jobs.size() - n_wgs follower items behind them
};

// Parameter offsets in the flat gradient vector = reference state_dict order (weight, bias per layer).
// mode: 0 = wgrad.hip (f32 MFMA), 1 = wgrad_split.hip (bf16x3), 2 = wgrad_f16.hip (f16x2): whose chunk times balance the slices
Plan make_plan(int ins_num, int64_t M, int max_wgs, int mode = 0) {
    const int C = ins_num + 1;
    const int64_t Mp = save_row_len(M);
    const int nchunks = (int)(Mp / KT);
    // row indices inside the SaveLayout buffers
    const int64_t R_pe = 0, R_de = POS_CH, R_h = POS_CH + DIR_CH, R_g1 = R_h + 8 * W, R_g2 = R_g1 + HW;
    // flat gradient offsets
    const Params PP = make_params(ins_num);
    int64_t w_m[8], b_m[8];
    for (int l = 0; l < 8; ++l) { w_m[l] = PP.mlps[l].w_off; b_m[l] = PP.mlps[l].b_off; }
    const int64_t w_rh = PP.rgb_hidden.w_off, b_rh = PP.rgb_hidden.b_off, b_ih = PP.ins_hidden.b_off;
    const int64_t w_d = PP.density.w_off, b_d = PP.density.b_off, w_io = PP.ins_out.w_off, b_io = PP.ins_out.b_off;
    const int64_t w_ro = PP.rgb_out.w_off, b_ro = PP.rgb_out.b_off;
    // scratch outputs at the head of the partials workspace: G = dg1 . h_7^T and Q = dg2 . h_7^T, [128][256] each (heads.hip)
    const int64_t S_G = 0, S_Q = HEAD_F_FLOATS;

    std::vector<JobDesc> d;
    // src ids: 0 = save (x), 1 = dsave (dy), 2 = transposed d raw [blk][4+C][32]
    const int GT = 4 + C;
    d.push_back({1, R_h + 0 * W, W, 0, 0, R_pe, POS_CH, 0, W, POS_CH, 0, w_m[0], POS_CH, 0, b_m[0]});
    for (int l = 1; l < 8; ++l) {
        const int ld = l == 5 ? W + POS_CH : W;
        d.push_back({1, R_h + (int64_t)l * W, W, 0, 0, R_h + (int64_t)(l - 1) * W, W, 0, W, W, 0, w_m[l], ld, 0, b_m[l]});
        if (l == 5) d.push_back({1, R_h + 5 * W, W, 0, 0, R_pe, POS_CH, 0, W, POS_CH, 0, w_m[5], ld, W, -1});   // cat[h, pts] (dm_nerf.py:87)
    }
    // the four linears around the activation-free feature layers: G and Q (scratch) + the two hidden bias gradients here,
    // d rgb_feature_linear(s), d ins_feature_linear(s) from them in head_unfuse_kernel
    // dg1 and dg2 are ONE 256-row tensor in the gradient workspace (layout.h::SaveLayout): [G ; Q] = [dg1 ; dg2] . h_7^T is one
    // fat 256 x 256 job that streams h_7 once (two 128 x 256 jobs read it twice: 256 of 5657 operand rows per chunk)
    static_assert(HEAD_F_FLOATS == HW * W, "G and Q are adjacent in the scratch head");
    const size_t i_G = d.size();
    {
        JobDesc gq{1, R_g1, 2 * HW, 0, 0, R_h + 7 * W, W, 0, 2 * HW, W, 0, S_G, W, 0, b_rh};
        gq.bias_split = HW; gq.bias_out_off2 = b_ih;
        d.push_back(gq);
        (void)S_Q;
    }
    d.push_back({1, R_g1, 2 * HW, 0, 0, R_de, DIR_CH, 0, HW, DIR_CH, 0, w_rh, W + DIR_CH, W, -1});             // cat[rgb_feature, dirs] (:90): rows 0..127 = dg1
    d.push_back({2, 0, GT, 3, 0, R_h + 7 * W, W, 0, 1, W, 0, w_d, W, 0, b_d});                                  // density_linear
    d.push_back({2, 0, GT, 4, 0, R_g2, HW, 0, C, HW, 0, w_io, HW, 0, b_io});                                   // ins_linear
    d.push_back({2, 0, GT, 0, 0, R_g1, HW, 0, 3, HW, 0, w_ro, HW, 0, b_ro});                                   // rgb_linear
    // Cost of one 32-sample chunk for a workgroup = its measured time in ns (scripts/diag_wgrad.py on MI355X,
    // r01): the MFMA time 256 * NBA * NBB cycles at ~2.4 GHz plus ~3 % for the fat classes; the skinny
    // classes are bound by the per-chunk hand-over (barrier + DMA issue), ~0.5 us.
    // split (wgrad_split.hip, six bf16 MFMAs per f32 product: 96 NBA NBB MFMA cycles per chunk): measured with
    // DMNERF_DIAG_SPLIT=1 scripts/diag_wgrad.py (r02q, re-measured r03x with the merged [dg1 ; dg2] job); the skinny classes keep
    // their hand-over / HBM floor.
    // f16x2 (wgrad_f16.hip, 48 NBA NBB MFMA cycles per chunk): every class sits on its operand stream; DMNERF_DIAG_SPLIT=f16
    // scripts/diag_wgrad.py (mean of r03w / r03x, both under this plan: profiles/r03/diag_wgrad_f16_r03w.txt, ..._r03x.txt; the classes ins_num 13 does not use: 174 ns per block).
    auto chunk_cost = [mode](int cls) {
        // (f32: re-fitted in r04 under the time-packed plan, every CU busy to the end of the launch -- scripts/diag_wgrad.py least
        // squares over the 256 workgroups: the 256 x 256 class 7307 ns at the clock the full chip sustains, the skinny classes
        // ~10 % above their r01 figures; only the RATIOS matter to the packing)
        static const double ns[N_CLASSES] = {/*8,8*/ 7307, /*4,8*/ 3680, /*8,2*/ 1950, /*4,1*/ 600, /*1,8*/ 1060,
                                             /*1,4*/ 600, /*2,4*/ 1100, /*3,4*/ 1500, /*4,4*/ 1900};
        // (both split tables re-fitted in r04 under the time-packed plan like the f32 one: the skinny classes +2 .. 6 %)
        static const double ns_split[N_CLASSES] = {/*8,8*/ 4600, /*4,8*/ 2700, /*8,2*/ 2030, /*4,1*/ 895, /*1,8*/ 1330,
                                                   /*1,4*/ 975, /*2,4*/ 1180, /*3,4*/ 1390, /*4,4*/ 1710};
        static const double ns_f16[N_CLASSES] = {/*8,8*/ 2811, /*4,8*/ 2100, /*8,2*/ 1341, /*4,1*/ 660, /*1,8*/ 950,
                                                 /*1,4*/ 738, /*2,4*/ 1043, /*3,4*/ 1217, /*4,4*/ 1391};
        return mode == 2 ? ns_f16[cls] : (mode == 1 ? ns_split[cls] : ns[cls]);
    };
    // Work items: every workgroup is filled to the same TIME T (wrap-around rule): walk the jobs in order, give the current
    // workgroup chunks of the current job until its budget is used, open the next workgroup, and when a job ends inside a
    // workgroup's budget let that workgroup go on with the first chunks of the next job.  The earlier plan gave every job an
    // integer number of equal slices, one per workgroup, so the workgroups of a job whose time share is 2.2 workgroups ran 3 (each
    // idle for a quarter of the launch) or 2 (the critical path): span 6.50 ms against 6.24 ms of divisible work (profiles/r03).
    // T is the smallest budget whose packing needs at most max_wgs workgroups (bisection); an item pays `item_ns` for its ring
    // fill and its partial-tile store on top of its chunks, and is never shorter than MINCH chunks unless the job is.
    for (auto& j : d) j.cls = class_for(j.rowsA, j.rowsB);
    struct Item { int job, chunk0, nchunk, lead; };        // lead: 1 = first item of its workgroup
    auto item_ns = [&](int cls) { return 1500.0 + 8.0 * (double)(CLS_NBA[cls] * 32 * CLS_NBB[cls] * 32) * 4.0 / 256.0; };   // ~1.5 us + the tile store
    constexpr int MINCH = 8;
    auto pack = [&](double T, std::vector<Item>* out) {
        int wgs = 0;
        double used = 0.0;            // time already in the current workgroup
        bool open = false;
        for (size_t jk = 0; jk < d.size(); ++jk) {
            const double c = chunk_cost(d[jk].cls), oh = item_ns(d[jk].cls);
            int c0 = 0, rem = nchunks;
            while (rem > 0) {
                if (!open) { ++wgs; used = 0.0; open = true; }
                int n = (int)std::floor((T - used - oh) / c);
                const int least = rem < MINCH ? rem : MINCH;
                if (n < least) {
                    if (used == 0.0) return 1 << 30;              // a fresh workgroup cannot hold the smallest item: T too small
                    open = false;                                 // not worth starting here: leave the rest of this budget idle
                    continue;
                }
                if (n > rem) n = rem;
                if (rem - n > 0 && rem - n < MINCH) n = rem - MINCH >= MINCH ? rem - MINCH : rem;      // no crumb for the next workgroup
                if (out) out->push_back({(int)jk, c0, n, used == 0.0 ? 1 : 0});
                used += oh + c * (double)n;
                c0 += n; rem -= n;
                if (T - used < 0.002 * T) open = false;
            }
        }
        return wgs;
    };
    double total_ns = 0.0;
    for (auto& j : d) total_ns += item_ns(j.cls) + chunk_cost(j.cls) * (double)nchunks;
    double lo = total_ns / (double)max_wgs, hi = total_ns * 1.01 + 1.0;   // hi: one workgroup takes everything (always feasible)
    for (int it = 0; it < 60 && hi - lo > 1e-4 * hi; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (pack(mid, nullptr) <= max_wgs) hi = mid; else lo = mid;
    }
    std::vector<Item> items;
    pack(hi, &items);
    std::vector<int> n_slices(d.size(), 0);
    for (const Item& it : items) ++n_slices[it.job];

    Plan P;
    P.part_floats = 2 * HEAD_F_FLOATS;                                       // G | Q first, the per-slice partials behind them
    for (size_t jk = 0; jk < d.size(); ++jk) {
        auto& j = d[jk];
        const int nba = CLS_NBA[j.cls], nbb = CLS_NBB[j.cls];
        const int ns = n_slices[jk];
        const int nshare = nbb >= 4 ? 4 : (nbb == 2 ? 2 : 1);               // Split<>::NSHARE
        const int64_t tile = (int64_t)nba * 32 * nbb * 32, brow = (int64_t)nba * 32 * nshare;
        WgOut o{};
        o.part_off = P.part_floats; o.slice_stride = tile + brow;
        o.bias_part_off = P.part_floats + tile; o.bias_slice_stride = tile + brow;
        o.out_off = j.out_off; o.bias_out_off = j.bias_out_off;
        o.n_slices = ns; o.rowsA = j.rowsA; o.rowsB = j.rowsB; o.ldp = nbb * 32; o.ld_out = j.ld_out; o.col_off = j.col_off;
        o.bias_sub = nshare; o.ldb = nba * 32;
        o.to_scratch = jk == i_G ? 1 : 0;
        o.bias_split = j.bias_split; o.bias_out_off2 = j.bias_out_off2;
        o.perm_a = j.a_src == 1;                                            // dy tensors of the dgrad pass
        o.perm_b = !(j.b_base == R_pe || j.b_base == R_de);                 // saved h / g1 / g2 (not the encodings)
        P.outs.push_back(o);
        for (int s = 0; s < ns; ++s) {
            const Item& it = items[P.jobs.size()];                          // (items are job-major, like this loop)
            WgJob g{};
            g.a_src = j.a_src; g.b_src = j.b_src;
            g.a_off = j.a_base * Mp; g.b_off = j.b_base * Mp;
            g.a_R = j.a_R; g.b_R = j.b_R; g.a_row0 = j.a_row0; g.b_row0 = j.b_row0;
            g.part_off = P.part_floats; g.bias_off = j.bias_out_off >= 0 ? P.part_floats + tile : -1;
            g.rowsA = j.rowsA; g.rowsB = j.rowsB; g.cls = j.cls;
            g.chunk0 = it.chunk0;
            g.nchunk = it.nchunk;
            g.follow = it.lead ? 0 : -1;
            P.jobs.push_back(g);
            P.part_floats += tile + brow;
        }
    }
    // The table: the leaders in workgroup order (= the grid), then the followers; a leader's followers are the items between it
    // and the next leader of the job-major sequence (the tail of one job, the head of the next), kept consecutive.
    {
        std::vector<WgJob> lead, foll;
        std::vector<int> first_foll;
        for (size_t i = 0; i < P.jobs.size(); ++i) {
            if (P.jobs[i].follow == 0) { lead.push_back(P.jobs[i]); first_foll.push_back(-1); }
            else {
                if (first_foll.back() < 0) first_foll.back() = (int)foll.size();
                ++lead.back().follow;
                foll.push_back(P.jobs[i]);
            }
        }
        for (size_t i = 0; i < lead.size(); ++i) lead[i].next = first_foll[i] < 0 ? -1 : (int)lead.size() + first_foll[i];
        for (auto& f : foll) f.next = -1;
        P.n_wgs = (int)lead.size();
        P.jobs = lead;
        P.jobs.insert(P.jobs.end(), foll.begin(), foll.end());
    }
    return P;
}

}  // namespace

static int plan_sizes(int mode, int ins_num, int64_t M, int max_wgs, int64_t* n_job_bytes, int64_t* n_out_bytes,
                      int64_t* part_floats, int* n_jobs, int* n_outs) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS || M < 1 || max_wgs < 32) return dmn_fail(DMNERF_E_ARG, "wgrad_plan: bad argument");
    const Plan P = make_plan(ins_num, M, max_wgs, mode);
    if (n_job_bytes) *n_job_bytes = (int64_t)(P.jobs.size() * sizeof(WgJob));
    if (n_out_bytes) *n_out_bytes = (int64_t)(P.outs.size() * sizeof(WgOut));
    if (part_floats) *part_floats = P.part_floats;
    if (n_jobs) *n_jobs = P.n_wgs;                    // the GRID: one block per workgroup; the table (n_job_bytes) also holds their follower items
    if (n_outs) *n_outs = (int)P.outs.size();
    return DMNERF_OK;
}

static int plan_fill(int mode, int ins_num, int64_t M, int max_wgs, void* h_jobs, int64_t job_bytes, void* h_outs, int64_t out_bytes) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS || M < 1 || max_wgs < 32 || !h_jobs || !h_outs) return dmn_fail(DMNERF_E_ARG, "wgrad_plan: bad argument");
    const Plan P = make_plan(ins_num, M, max_wgs, mode);
    if (job_bytes != (int64_t)(P.jobs.size() * sizeof(WgJob)) || out_bytes != (int64_t)(P.outs.size() * sizeof(WgOut)))
        return dmn_fail(DMNERF_E_ARG, "wgrad_plan: buffer sizes do not match dmnerf_wgrad_plan_sizes");
    memcpy(h_jobs, P.jobs.data(), (size_t)job_bytes);
    memcpy(h_outs, P.outs.data(), (size_t)out_bytes);
    return DMNERF_OK;
}

extern "C" int dmnerf_wgrad_plan_sizes(int ins_num, int64_t M, int max_wgs, int64_t* n_job_bytes, int64_t* n_out_bytes,
                                       int64_t* part_floats, int* n_jobs, int* n_outs) {
    return plan_sizes(0, ins_num, M, max_wgs, n_job_bytes, n_out_bytes, part_floats, n_jobs, n_outs);
}
extern "C" int dmnerf_wgrad_plan(int ins_num, int64_t M, int max_wgs, void* h_jobs, int64_t job_bytes, void* h_outs, int64_t out_bytes) {
    return plan_fill(0, ins_num, M, max_wgs, h_jobs, job_bytes, h_outs, out_bytes);
}
// the same tables balanced for the chunk times of the split-bf16 kernel (wgrad_split.hip)
extern "C" int dmnerf_wgrad_plan_sizes_split(int ins_num, int64_t M, int max_wgs, int64_t* n_job_bytes, int64_t* n_out_bytes,
                                             int64_t* part_floats, int* n_jobs, int* n_outs) {
    return plan_sizes(1, ins_num, M, max_wgs, n_job_bytes, n_out_bytes, part_floats, n_jobs, n_outs);
}
extern "C" int dmnerf_wgrad_plan_split(int ins_num, int64_t M, int max_wgs, void* h_jobs, int64_t job_bytes, void* h_outs, int64_t out_bytes) {
    return plan_fill(1, ins_num, M, max_wgs, h_jobs, job_bytes, h_outs, out_bytes);
}
// ... and for those of the split-f16 kernel (wgrad_f16.hip)
extern "C" int dmnerf_wgrad_plan_sizes_f16(int ins_num, int64_t M, int max_wgs, int64_t* n_job_bytes, int64_t* n_out_bytes,
                                           int64_t* part_floats, int* n_jobs, int* n_outs) {
    return plan_sizes(2, ins_num, M, max_wgs, n_job_bytes, n_out_bytes, part_floats, n_jobs, n_outs);
}
extern "C" int dmnerf_wgrad_plan_f16(int ins_num, int64_t M, int max_wgs, void* h_jobs, int64_t job_bytes, void* h_outs, int64_t out_bytes) {
    return plan_fill(2, ins_num, M, max_wgs, h_jobs, job_bytes, h_outs, out_bytes);
}

long long* g_dmn_wgrad_trace = nullptr;      // (wgrad_split.hip shares the diagnostic hook)
extern "C" int dmnerf_wgrad_set_trace(int64_t* d_ticks) {
    g_dmn_wgrad_trace = (long long*)d_ticks;
    return DMNERF_OK;
}

extern "C" int dmnerf_mlp_bwd_weights(const float* d_save, const float* d_dsave, const float* d_graw_t, int64_t M,
                                      const void* d_jobs, int n_jobs, const void* d_outs, int n_outs,
                                      const float* d_params_flat, int ins_num, float* d_part, float* d_grad_flat, void* stream) {
    if (!d_save || !d_dsave || !d_graw_t || !d_jobs || !d_outs || !d_params_flat || !d_part || !d_grad_flat || M < 1 || n_jobs < 1 || n_outs < 1)
        return dmn_fail(DMNERF_E_ARG, "mlp_bwd_weights: bad argument");
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_weights: ins_num %d unsupported", ins_num);
    WgArgs a{};
    a.src[0] = d_save; a.src[1] = d_dsave; a.src[2] = d_graw_t;
    a.part = d_part; a.jobs = (const WgJob*)d_jobs; a.Mp = save_row_len(M); a.trace = g_dmn_wgrad_trace;
    const size_t lds_bytes = WG_LDS_BYTES;
    static DmnOncePerDevice once;
    if (hipError_t e = once.run([&] { return hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); });
        e != hipSuccess)
        return dmn_fail_hip(e, "mlp_bwd_weights: hipFuncSetAttribute");
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)n_jobs), dim3(256), lds_bytes, (hipStream_t)stream, a);
    int rc = dmn_check_launch("mlp_bwd_weights");
    if (rc) return rc;
    return dmn_wgrad_finish(d_outs, n_outs, d_params_flat, ins_num, d_part, d_grad_flat, (hipStream_t)stream);
}

int dmn_wgrad_finish(const void* d_outs, int n_outs, const float* d_params_flat, int ins_num, float* d_part, float* d_grad_flat, hipStream_t stream,
                     const float* d_unscale) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(64, (unsigned)n_outs), dim3(256), 0, stream, d_part, (const WgOut*)d_outs, n_outs, d_grad_flat, d_unscale);
    const int rc = dmn_check_launch("mlp_bwd_weights: reduce");
    if (rc) return rc;
    // rgb_feature_linear(s) / ins_feature_linear(s) from G, Q and the hidden layers' bias gradients (heads.hip)
    return dmn_head_unfuse(d_params_flat, ins_num, d_part, d_part + HEAD_F_FLOATS, d_grad_flat, stream);
}
