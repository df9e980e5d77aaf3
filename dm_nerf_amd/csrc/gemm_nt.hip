// gemm_nt.hip -- the GEMM of the layer-by-layer DM_NeRF path (network shapes other than the shipped D = 8 / W = 256; generic.py)
// on the machine's own terms: operands global -> LDS by LDS-DMA, ds_read_b128 operand reads, hand-scheduled MFMA stream.
//
//   C[m][n] = act( sum_k A[m][k] W[n][k] + bias[n] )        A: activations [M rows of samples], W: packed weights, both k-fast
//
// * forward of a linear layer:  A = the layer's input rows, W = nn.Linear.weight (dm_nerf.py:80-106);
// * data gradient:              A = dL/dy rows, W = weight^T (packed transposed), epilogue . [h > 0] and optional accumulate;
// * the cats of dm_nerf.py:87,90 ([h, pts], [rgb_feature, dirs]) are a K-OFFSET, not a copy: the A operand has up to two K ranges
//   with their own source pointer and row stride (h | pts), matched by the column order of the packed weights.
//
// Tile: one workgroup = 128 samples x ALL (up to 320) outputs of the layer: wave w owns the 32 samples of A block w and every
// B block, NBB accumulator blocks of v_mfma_f32_32x32x2_f32 (the weights are the shared operand: one copy per workgroup through
// LDS; each activation row is read from HBM exactly once per layer).  Per 32-k chunk the tile is (4 + NBB) blocks of 32 rows x
// 128 bytes; a block lands in LDS by four LDS-DMA wave-instructions (buffer_load ... lds, 16 B per lane: 8 rows of 128 B each) with the 16-byte units of a
// row XOR-swizzled by (row >> 1) & 7, so that the ds_read_b128 of 32 different rows is conflict-free (the geometry of wgrad.hip's
// ring, with row strides instead of contiguous tiles).  A D-deep ring (2 .. 4 chunks) keeps the next chunks in flight; the hand-over
// (vmcnt + barrier) sits at the start of a chunk's last round; operand reads run one round ahead through inline asm, one per MFMA
// gap; the refills of the released slot are spread over the last round's gaps.  Per round: 1 + NBB reads for 4 NBB MFMAs.
// Exact f32 (the MFMA is an fmaf chain), bias as the accumulator's initial value, no vendor BLAS.
//
// Roofline: MFMA f32 for W >= 192 (2 N K flop per 4 (K + N) bytes of HBM per sample); for narrower layers the HBM round trip of the
// activations between two launches is the bound.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "mlp_common.h"

using namespace dmn;

namespace {

constexpr int NT_STAGE_BYTES = 16384;     // the epilogue's transposition area: 4 KiB per wave (one 32 x 32 block)
// (ring budget: 147 456 bytes of the CU's 160 KiB when ONE workgroup owns the CU, 81 920 when TWO share it, minus the staging area)
constexpr int NT_MAX_DEPTH = 4;
constexpr int NT_MAX_NBB = 10;            // 320 outputs per workgroup: 160 accumulator registers (12 blocks no longer fit the file next to the epilogue's staging)

// Workgroups per CU.  Up to 6 out-blocks a wave needs < 256 registers and the ring fits 80 KiB at depth 2: two workgroups share a CU
// and one's epilogue hides under the other's MFMA stream (W = 192: 0.66 of the roof against 0.64 alone on the CU with the 16-byte
// epilogue, measured); wider tiles keep the CU.  The epilogue's staging area fits next to the ring except at 5 and 6 out-blocks with
// two workgroups (2 x 40 KiB + 16 > 80): those keep the dword epilogue.
constexpr int nt_occupancy(int nbb) { return nbb <= 6 ? 2 : 1; }
constexpr bool nt_staged(int nbb) { return nbb <= 4 || nbb > 6; }

struct NtArgs {
    const float* A0; const float* A1;     // the two K ranges of the A operand (A1 null: one range)
    int64_t lda0, lda1;                   // row strides (floats, multiples of 4: rows are 16-byte aligned)
    int64_t a0_floats, a1_floats;         // floats from A0 / A1 to the end of their allocations
    int nc0, nc1;                         // 32-k chunks of each range
    const float* B; int64_t b_floats;     // packed weights [rows padded to 32 NBB x tiles][ldb], zero-filled padding
    int ldb;                              // = 32 (nc0 + nc1)
    const float* bias;                    // [padded rows] or null
    float* C; int64_t ldc;                // C[m * ldc + n]
    int n_store, n_zero;                  // columns [0, n_store) get values, [n_store, n_zero) zeros (the pad columns of a row)
    const float* mask; int64_t ldm;       // data gradient: C = mask[m * ldm + n] > 0 ? v : 0, or null
    int64_t M;
    int relu, accumulate;
    int x4;                               // epilogue form: 1 = 16-byte stores through the LDS transposition (n_zero a multiple of 4)
#ifdef DMN_NT_TRACE
    long long* trace;                     // diagnostic builds only (scripts/diag_gemm_nt.py): per workgroup and tile, shader-clock stamps
#endif
};
#ifdef DMN_NT_TRACE
#define DMN_NT_STAMP(k) do { if (a.trace && threadIdx.x == 0 && tile_seq < 32) a.trace[((int64_t)blockIdx.x * 32 + tile_seq) * 4 + (k)] = (long long)clock64(); } while (0)
#else
#define DMN_NT_STAMP(k) do {} while (0)
#endif

template <int NBB>
struct NtRing {
    static constexpr int NL = 4 + NBB;                                   // DMA pieces per wave per chunk (1 KiB each)
    static constexpr int BUF = NL * 4096;                                // bytes per chunk
    static constexpr int STAGE = nt_staged(NBB) ? NT_STAGE_BYTES : 0;
    static constexpr int BUDGET = (nt_occupancy(NBB) == 2 ? 81920 : 147456) - STAGE;
    static constexpr int D = BUDGET / BUF < NT_MAX_DEPTH ? BUDGET / BUF : NT_MAX_DEPTH;
    static_assert(D >= 2, "ring needs two slots");
    static_assert((D - 1) * NL <= 63, "vmcnt range");
};

template <int NBB>
__global__ __launch_bounds__(256, nt_occupancy(NBB)) void gemm_nt_kernel(const NtArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)     // (the HOST pass only needs the launch stub: it instantiates an empty body -- the device-only builtins and
                                        //  register constraints below made its instantiation of the full body fail silently, and with it the stub)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef NtRing<NBB> RG;
    constexpr int NL = RG::NL, D = RG::D, BUF = RG::BUF;
    constexpr int NR = 1 + NBB;                     // operand reads per round
    constexpr int NGAP = 4 * NBB;                   // MFMAs per round
    const int tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = lds_addr(lds);
    const int j0 = blockIdx.y * (NBB * 32);         // first output (row of the packed weights) of this workgroup
    const int nchunk = a.nc0 + a.nc1;
    // PERSISTENT: the grid is (at most) one workgroup per CU slot and a workgroup walks the sample tiles blockIdx.x, + gridDim.x, ...
    // (a tile's fixed cost -- workgroup launch, the first chunks' DMA latency, the stores draining -- was ~10 us against 5 .. 40 us of MFMA work)
    // (32-bit tile counters: a 64-bit `<` has no scalar form -- it compiled to a vector compare, the fetch state moved to VGPRs with it and
    // every DMA request sat in a waterfall loop; the host bounds M)
    const int ntiles = (int)((a.M + 127) / 128);
    int tile = blockIdx.x;

    // ---- DMA geometry (wgrad.hip): wave w owns the 1-KiB piece w of every 32-row block (rows 8 w .. 8 w + 7); lane l lands at LDS
    // row 8 w + (l >> 3), unit l & 7, so it must FETCH unit (l & 7) ^ ((row >> 1) & 7) of that row
    const int drow = 8 * w + (lane >> 3);
    const int dunit = ((lane & 7) ^ ((drow >> 1) & 7)) << 4;
    const int voA0 = (int)(drow * a.lda0 * 4) + dunit;
    const int voA1 = (int)(drow * a.lda1 * 4) + dunit;
    const int voB = drow * a.ldb * 4 + dunit;
    auto bound = [](int64_t want, int64_t have) { const int64_t b = want < have ? want : have; return b < 0x1fffffff ? b : (int64_t)0x1fffffff; };
    const rsrc_t rsB = uniform_rsrc(a.B + (int64_t)j0 * a.ldb, bound((int64_t)NBB * 32 * a.ldb, a.b_floats - (int64_t)j0 * a.ldb));
    const int blkA0 = (int)(32 * a.lda0 * 4), blkA1 = (int)(32 * a.lda1 * 4), blkB = 32 * a.ldb * 4;      // bytes per 32-row block
    // `args()`: the kernel arguments re-read from the kernarg segment where a tile's set-up / epilogue needs them.  Read once at the
    // top they are ~60 SGPRs held across the whole tile loop (the compiler hoists the loads), and the loop's own state spilled.
    typedef const NtArgs __attribute__((address_space(4))) KArgs;         // (the kernel's one argument sits at the start of the segment)
    auto args = [&]() -> KArgs* { KArgs* p = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(p)); return p; };
    // per-tile state: rows beyond M are beyond the A descriptors' ranges (they read as 0 and feed rows the epilogue never stores)
    int64_t i0 = 0, rows_valid = 0;
    rsrc_t rsA0, rsA1;
    auto set_tile = [&](int t) __attribute__((always_inline)) {
        KArgs* q = args();
        i0 = (int64_t)t * 128;
        rows_valid = q->M - i0 < 128 ? q->M - i0 : 128;
        rsA0 = uniform_rsrc(q->A0 + i0 * q->lda0, bound(rows_valid * q->lda0, q->a0_floats - i0 * q->lda0));
        rsA1 = q->A1 ? uniform_rsrc(q->A1 + i0 * q->lda1, bound(rows_valid * q->lda1, q->a1_floats - i0 * q->lda1)) : rsA0;
    };

    // (`fresh_s`: the block strides are re-read where they are used -- as loop invariants of the tile loop the 4 + NBB piece offsets,
    // times two call sites, were a hundred SGPRs held across the whole kernel and spilled)
    auto fresh_s = [](int x) -> int { asm volatile("" : "+s"(x)); return x; };
    auto dma_chunk_piece = [&](int cc, unsigned slot_byte, int i) __attribute__((always_inline)) {      // piece i of NL of chunk cc (< nchunk) into a ring slot
        float* dst = lds + (slot_byte + i * 4096 + fresh_s(w) * 1024) / 4;       // (likewise: the LDS addresses of the first chunks' pieces)
#if defined(DMN_NT_NO_A)   /* diagnostic builds only (scripts/diag_gemm_nt.sh): the loop without the activation / weight requests */
        if (i < 4) return;
#endif
#if defined(DMN_NT_NO_B)
        if (i >= 4) return;
#endif
        if (i < 4) {
            if (cc < a.nc0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA0, (DMN_LAS void*)dst, 16, voA0, i * fresh_s(blkA0) + cc * 128, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA1, (DMN_LAS void*)dst, 16, voA1, i * fresh_s(blkA1) + (cc - a.nc0) * 128, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (DMN_LAS void*)dst, 16, voB, (i - 4) * fresh_s(blkB) + cc * 128, 0, 0);
        }
    };
    // ---- read geometry: lane (li, half) reads row 32 blk + li, unit (2 t + half) ^ ((li >> 1) & 7) in round t
    unsigned offA[4], offB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned o = lds0 + li * 128 + ((((2 * t + half) ^ ((li >> 1) & 7))) << 4);
        offA[t] = o + w * 4096;
        offB[t] = o + 4 * 4096;
    }

    // ---- the bias (column n = j0 + 32 b + li is this lane's in every register of block b), once for all tiles
    float bias_v[NBB];
#pragma unroll
    for (int b = 0; b < NBB; ++b) {
        bias_v[b] = a.bias ? a.bias[j0 + 32 * b + li] : 0.f;
        if constexpr (NBB > 6) asm volatile("" : "+a"(bias_v[b]));          // (parked in the AGPR half next to the accumulators)
    }
    asm volatile("" ::: "memory");

    f32x4 av[2][1], bv[2][NBB];
    auto read_ops_one = [&](auto gc, int buf, unsigned addrA, unsigned addrB) {     // operand g of a round
        constexpr int g = decltype(gc)::value;
        if constexpr (g == 0) lds_read16_async<0>(av[buf][0], addrA);
        else lds_read16_async<(g - 1) * 4096>(bv[buf][g - 1], addrB);
    };

    // ---- ONE chunk stream across the tiles.  The ring does not drain at a tile boundary: the chunk fetched D chunks ahead simply
    // belongs to the next tile once the current one's K range is exhausted (its own A descriptors, the weights again from k = 0), so a
    // tile's first chunks have long landed when its first MFMA issues, and the memory latency is paid once per workgroup, not per tile.
    // Fetch state: the tile / chunk the NEXT request is for (rsA0 / rsA1 are the FETCH tile's descriptors); compute state: tile, i0.
    int fc = 0;                                         // chunk (inside the fetch tile) of the next request
    int ahead = 0;                                      // chunks requested beyond the one being consumed
    int ftile = tile;
    bool fvalid = true;
    set_tile(ftile);
    const int64_t i0_first = i0;
    auto advance_fetch = [&]() __attribute__((always_inline)) {           // after the NL pieces of (ftile, fc) have been issued
        if (++fc == nchunk) {
            fc = 0;
            ftile += (int)gridDim.x;
            fvalid = ftile < ntiles;
            if (fvalid) set_tile(ftile);
        }
    };
    // prologue: the first D chunks of the stream (they may span tiles when K is short), chunk 0 landed, its round-0 operands on their way
#pragma unroll
    for (int sl = 0; sl < D; ++sl)
        if (fvalid) {
#pragma unroll
            for (int i = 0; i < NL; ++i) dma_chunk_piece(fc, sl * BUF, i);
            advance_fetch();
            ++ahead;
        }
    --ahead;                                            // (chunk 0 is the one being consumed)
    if (ahead == D - 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 1) * NL) & 15) | ((((D - 1) * NL) >> 4) << 14));     // vmcnt((D-1) NL) only
    else __builtin_amdgcn_s_waitcnt(0x0F70);                                                                           // a short stream: everything
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    static_for<NR>([&](auto gc) { read_ops_one(gc, 0, offA[0], offB[0]); });

    f32x16 acc[NBB];
    unsigned sb = 0;                                    // byte offset of the ring slot of the chunk being consumed (uniform)
    int64_t i0c = i0_first;                             // the tile being COMPUTED
#ifdef DMN_NT_TRACE
    int tile_seq = 0;
#endif
#pragma nounroll
    for (;;) {
        DMN_NT_STAMP(0);
#pragma unroll
        for (int b = 0; b < NBB; ++b) {                 // accumulators start from the bias
            float bb = bias_v[b];
            asm volatile("" : "+v"(bb));                // (opaque per trip: the 16-wide splats are loop invariants otherwise -- 16 NBB registers
            acc[b] = (f32x16)(bb);                      //  held across the tile loop next to the accumulators themselves)
            if constexpr (NBB > 6) asm volatile("" : "+a"(acc[b]));    // one workgroup per CU: the accumulators live in the AGPR half of the file
        }

#pragma nounroll
        for (int c = 0; c < nchunk; ++c) {
            const unsigned nb = sb + BUF == (unsigned)(D * BUF) ? 0u : sb + BUF;
            unsigned cA[4], cB[4];
#pragma unroll
            for (int t = 1; t < 4; ++t) { cA[t] = offA[t] + sb; cB[t] = offB[t] + sb; }
            cA[0] = offA[0] + nb; cB[0] = offB[0] + nb; // round 0 of the NEXT chunk of the stream (read in this chunk's round 3)
            static_for<4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                lds_wait<0>(av[r & 1]);
#pragma unroll
                for (int k = 0; k < NBB; ++k) asm volatile("" : "+" DMN_TILE_RC(bv[r & 1][k]));
                if constexpr (r == 3) {
                    // ring hand-over: the next chunk of the stream has landed in every wave's view, and this chunk's slot is released.
                    // In flight behind the next chunk are the D - 2 chunks after it (plus, after a tile boundary, the epilogue's stores:
                    // younger, so the count only errs towards waiting) -- or fewer at the end of the stream, where everything is waited for
                    if (ahead == D - 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 2) * NL) & 15) | ((((D - 2) * NL) >> 4) << 14));
                    else __builtin_amdgcn_s_waitcnt(0x0F70);
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                static_for<NGAP>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    constexpr int u = g / NBB, ib = g % NBB;
                    if constexpr (g < NR) read_ops_one(gc, (r + 1) & 1, cA[(r + 1) & 3], cB[(r + 1) & 3]);
                    if constexpr (r == 3) {                             // refill the released slot with the stream's next chunk (if any)
                        constexpr int G0 = NR < NGAP ? NR : NGAP - 1;
                        constexpr int PD = (NGAP - G0) / NL > 0 ? (NGAP - G0) / NL : 1;
                        static_for<NL>([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            constexpr int at = G0 + i * PD < NGAP ? G0 + i * PD : NGAP - 1;
                            if constexpr (at == g) {
                                if (fvalid) dma_chunk_piece(fc, sb, i);
                            }
                        });
                    }
                    acc[ib] = mfma32(av[r & 1][0][u], bv[r & 1][ib][u], acc[ib]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            // this chunk is consumed (ahead - 1); its slot took the stream's next chunk, if there was one (ahead + 1)
            if (fvalid) advance_fetch();
            else --ahead;
            sb = nb;
        }
        // the round-0 operands of the stream's next chunk (the NEXT tile's first) were requested in the last round: landed before the
        // epilogue's own memory operations start (the ties hand the registers back to the compiler)
        DMN_NT_STAMP(1);
        lds_wait<0>(av[0]);
#pragma unroll
        for (int k = 0; k < NBB; ++k) asm volatile("" : "+" DMN_TILE_RC(bv[0][k]));

        const int64_t i0_done = i0c;
        const int64_t rows_done = args()->M - i0_done < 128 ? args()->M - i0_done : 128;
        const int next = tile + (int)gridDim.x;
        const bool more = next < ntiles;
        i0c = (int64_t)next * 128;

        // ---- epilogue: lane holds column n = j0 + 32 b + li, rows 32 w + (r & 3) + 8 (r >> 2) + 4 half
        // (`fresh` hides a value's origin: what is derived from it is computed HERE, in every trip, instead of being hoisted out of the
        // tile loop -- 16 NBB store offsets as loop invariants were 130 spilled SGPRs)
        auto fresh_v = [](int x) -> int { asm volatile("" : "+v"(x)); return x; };
        KArgs* q = args();
        const int ncols = q->n_zero - j0 < NBB * 32 ? q->n_zero - j0 : NBB * 32;       // columns of this workgroup that exist in C
        float* const Ct = q->C + i0_done * q->ldc + j0;
        const rsrc_t rsC = uniform_rsrc(Ct, (rows_done - 1) * q->ldc + ncols);    // rows beyond M fall outside: dropped by the hardware
        const int lane_e = fresh_v(lane);
        const int half_e = lane_e >> 5, li_e = lane_e & 31;
        const int voC = (int)(((int64_t)(32 * w + 4 * half_e) * q->ldc + li_e) * 4);
        const int rowB = fresh_v((int)(q->ldc * 4));                               // (in a VGPR: the 16 row offsets of a block are VALU adds here, not 16 SGPRs)
        rsrc_t rsM = rsC;
        int voM = 0, rowM = 0;
        if (q->mask) {
            rsM = uniform_rsrc(q->mask + i0_done * q->ldm + j0, (rows_done - 1) * q->ldm + ncols);
            voM = (int)(((int64_t)(32 * w + 4 * half_e) * q->ldm + li_e) * 4);
            rowM = fresh_v((int)(q->ldm * 4));
        }
        if (nt_staged(NBB) && q->x4) {
            // ---- 16-byte stores.  A burst of 16 NBB dword stores per lane ran into the wave's limit of outstanding memory operations:
            // 15 000 cycles per tile at 10 out-blocks, 36 000 with two workgroups per CU (stamps: scripts/diag_gemm_nt.py) -- a sixth to a
            // third of the tile.  Each 32 x 32 block goes through this wave's 4 KiB of LDS instead (lane (li, half) writes column li of
            // its 16 rows; lane l reads 4 consecutive columns of row 8 j + (l >> 3)) and leaves as FOUR 1-KiB stores: 8 rows x 128 bytes.
            float* const st = lds + (D * BUF) / 4 + w * 1024;
            const int row_l = lane_e >> 3, col_l = 4 * (lane_e & 7);
            const int voC4 = (int)(((int64_t)(32 * w + row_l) * q->ldc + col_l) * 4);
            const int voM4 = q->mask ? (int)(((int64_t)(32 * w + row_l) * q->ldm + col_l) * 4) : 0;
            const int rowB8 = 8 * rowB, rowM8 = 8 * rowM;
            const bool all_valid = j0 + NBB * 32 <= q->n_store;
#pragma unroll
            for (int b = 0; b < NBB; ++b) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[((r & 3) + 8 * (r >> 2) + 4 * half_e) * 32 + li_e] = acc[b][r];
                const int col0 = 32 * b + col_l;
                const bool in_c = col0 < ncols;                                  // (ncols is a multiple of 4 here)
                const int vo = in_c ? voC4 + b * 128 : 0x7ffffff0;
                const int vm = in_c ? voM4 + b * 128 : 0x7ffffff0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(st + (8 * j + row_l) * 32 + col_l);
                    if (q->accumulate) {
                        const u32x4 o = __builtin_amdgcn_raw_buffer_load_b128(rsC, vo + j * rowB8, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += __uint_as_float(o[e]);
                    }
                    if (q->relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = relu1(v[e]);
                    }
                    if (q->mask) {
                        const u32x4 mk = __builtin_amdgcn_raw_buffer_load_b128(rsM, vm + j * rowM8, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(mk[e]) > 0.f ? v[e] : 0.f;
                    }
                    u32x4 o;
                    if (all_valid) {                                            // (uniform: no pad column inside this workgroup's columns)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = f2u(v[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = j0 + col0 + e < q->n_store ? f2u(v[e]) : 0u;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o, rsC, vo + j * rowB8, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
    #pragma unroll
            for (int b = 0; b < NBB; ++b) {
                const int col = 32 * b + li_e;
                const bool in_c = col < ncols;                                       // (a column predicate: the descriptor bounds rows only)
                const bool is_val = j0 + col < q->n_store;
                const int vo = in_c ? voC + b * 128 : 0x7ffffff0;
                const int vm = in_c ? voM + b * 128 : 0x7ffffff0;
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    float v = acc[b][r];
                    if (q->accumulate) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsC, vo + ro * rowB, 0, 0));
                    if (q->relu) v = relu1(v);
                    if (q->mask) v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsM, vm + ro * rowM, 0, 0)) > 0.f ? v : 0.f;
                    v = is_val ? v : 0.f;
    #if defined(DMN_NT_NO_STORE)   /* diagnostic: only the first block's stores (keeps the epilogue's arithmetic alive) */
                    if (b > 0) { asm volatile("" :: "v"(v)); continue; }
    #endif
                    __builtin_amdgcn_raw_buffer_store_b32(f2u(v), rsC, vo + ro * rowB, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);           // block by block: the scheduler would otherwise pull every accumulator out of the
            }                                                // AGPR file first (16 NBB VGPRs live at once: spills in the wide tiles)
        }
        DMN_NT_STAMP(2);
#ifdef DMN_NT_TRACE
        ++tile_seq;
#endif
        if (!more) break;
        tile = next;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
    (void)a;
#endif
}

// Packed weights of one layer: out[n][kk] for n < rows_pad, kk < ldb -- range 0 = source columns [c0, c0 + k0) at kk = 0 .., range 1 =
// source columns [c1, c1 + k1) at kk = 32 ceil(k0 / 32) ..; everything else zero.  transposed: the source is read as W^T
// (out[n][kk] = W[kk-th source ROW][n-th source COLUMN]) -- the data-gradient form.  bias_out[n] = bias[n] (zero beyond N).
struct PackNtArgs {
    const float* W; int64_t ldw; int N;           // source [*, ldw]; N = logical rows of the packed matrix
    int c0, k0, c1, k1;
    int transposed;
    float* out; int rows_pad, ldb;
    const float* bias; float* bias_out;
};

__global__ void pack_nt_kernel(const PackNtArgs a) {
    const int64_t total = (int64_t)a.rows_pad * a.ldb;
    const int kp0 = (a.k0 + 31) / 32 * 32;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(e / a.ldb), kk = (int)(e % a.ldb);
        int src = -1;
        if (kk < a.k0) src = a.c0 + kk;
        else if (kk >= kp0 && kk - kp0 < a.k1) src = a.c1 + (kk - kp0);
        float v = 0.f;
        if (n < a.N && src >= 0) v = a.transposed ? a.W[(int64_t)src * a.ldw + n] : a.W[(int64_t)n * a.ldw + src];
        a.out[e] = v;
    }
    if (a.bias_out)
        for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < a.rows_pad; n += gridDim.x * blockDim.x)
            a.bias_out[n] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
}

// pts = o + d z, viewdirs = d / |d| per sample (render.py:37,49-57) and both positional encodings (Embedder.embed, dm_nerf.py:37-38)
// straight into row-padded buffers: x_pos [M][ldp], x_dir [M][ldv] with the pad columns zeroed -- the A operands of gemm_nt
// (rows 16-byte aligned).  A workgroup owns 64 consecutive samples: one thread per (sample, coordinate) computes into an LDS tile (the
// same shared range reduction as the fused kernels), then the whole workgroup writes the tile's rows -- one contiguous range of each
// buffer -- as 16-byte stores.  (Written per thread, a wave's store touched 21 rows x 12 bytes: 145 us for 0.3 GB, a third of the HBM rate.)
constexpr int RE_ROWS = 64;
__global__ __launch_bounds__(256) void ray_embed_kernel(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ z,
                                                        int64_t N, int S, int Lp, int Lv, float* __restrict__ xp, int ldp, float* __restrict__ xv,
                                                        int ldv, int vec) {
    extern __shared__ __attribute__((aligned(16))) float re_tile[];
    const int sp = ldp + 4, sv = ldv + 4;                          // (row strides off the 32-bank period; rows stay 16-byte aligned)
    float* const tp = re_tile;
    float* const tv = re_tile + RE_ROWS * sp;
    const int64_t M = N * S;
    const int64_t m0 = (int64_t)blockIdx.x * RE_ROWS;
    const int t = threadIdx.x;
    if (t < 3 * RE_ROWS && m0 + t / 3 < M) {
        const int c = t % 3;
        const int64_t m = m0 + t / 3;
        const int64_t n = m / S;
        const float dx = rd[n * 3], dy = rd[n * 3 + 1], dz = rd[n * 3 + 2], zv = z[m];
        const float dc = c == 0 ? dx : (c == 1 ? dy : dz);
        const float p = ro[n * 3 + c] + dc * zv;                   // render.py:49: separate multiply and add
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);      // render.py:37
        const float v = dc / nrm;
        float* op = tp + (t / 3) * sp;
        float* ov = tv + (t / 3) * sv;
        op[c] = p;
        ov[c] = v;
        // one sin / cos pair per coordinate (frequency 1, shared range reduction in double), the higher octaves by the double-angle
        // recurrence in double precision -- exactly what the fused kernels' encode() does (mlp_common.h): 5 instead of ~19 f64 instructions
        // per output, absolute error <= 2^k 1e-16
        {
            const double tt = dmn::rev_of(p);
            double sn = dmn::sin_rev_d(tt, 0, 0), cs = dmn::sin_rev_d(tt, 0, 1);
            for (int k = 0; k < Lp; ++k) {
                op[3 + 6 * k + c] = (float)sn;
                op[3 + 6 * k + 3 + c] = (float)cs;
                const double s_old = sn, t2 = s_old + s_old;
                sn = t2 * cs;
                cs = __builtin_fma(-t2, s_old, 1.0);
            }
        }
        {
            const double tt = dmn::rev_of(v);
            double sn = dmn::sin_rev_d(tt, 0, 0), cs = dmn::sin_rev_d(tt, 0, 1);
            for (int k = 0; k < Lv; ++k) {
                ov[3 + 6 * k + c] = (float)sn;
                ov[3 + 6 * k + 3 + c] = (float)cs;
                const double s_old = sn, t2 = s_old + s_old;
                sn = t2 * cs;
                cs = __builtin_fma(-t2, s_old, 1.0);
            }
        }
        for (int q = 3 + 6 * Lp + c; q < ldp; q += 3) op[q] = 0.f;     // pad columns
        for (int q = 3 + 6 * Lv + c; q < ldv; q += 3) ov[q] = 0.f;
    }
    __syncthreads();
    const int rows = (int)(M - m0 < RE_ROWS ? M - m0 : RE_ROWS);
    float* const gp = xp + m0 * ldp;
    float* const gv = xv + m0 * ldv;
    if (vec) {                                                     // rows and bases 16-byte aligned (the host checked)
        const int qp = ldp >> 2, qv = ldv >> 2;
        for (int e = t; e < rows * qp; e += 256) {
            const int r = e / qp, q = e - r * qp;
            reinterpret_cast<f32x4*>(gp)[e] = *reinterpret_cast<const f32x4*>(tp + r * sp + 4 * q);
        }
        for (int e = t; e < rows * qv; e += 256) {
            const int r = e / qv, q = e - r * qv;
            reinterpret_cast<f32x4*>(gv)[e] = *reinterpret_cast<const f32x4*>(tv + r * sv + 4 * q);
        }
    } else {
        for (int e = t; e < rows * ldp; e += 256) gp[e] = tp[(e / ldp) * sp + e % ldp];
        for (int e = t; e < rows * ldv; e += 256) gv[e] = tv[(e / ldv) * sv + e % ldv];
    }
}

// dst[m][c] = src[m][c] for c < n, 0 for n <= c < n_pad: a column slice as a row-padded gemm_nt operand
__global__ void copy_cols_pad_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd, int64_t M, int n, int n_pad) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * n_pad) return;
    const int64_t m = e / n_pad;
    const int c = (int)(e % n_pad);
    dst[m * ldd + c] = c < n ? src[m * lds_ + c] : 0.f;
}

template <int NBB>
int launch_nt(const NtArgs& a, int tiles_n, hipStream_t stream) {
    constexpr int lds_bytes = NtRing<NBB>::D * NtRing<NBB>::BUF + NtRing<NBB>::STAGE;
    static DmnOncePerDevice once;
    if (hipError_t e = once.run([] { return hipFuncSetAttribute((const void*)gemm_nt_kernel<NBB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); });
        e != hipSuccess)
        return dmn_fail_hip(e, "gemm_nt: hipFuncSetAttribute");
    // persistent grid: one workgroup per CU slot (nt_occupancy per CU), each walking the sample tiles with stride gridDim.x
    int dev = 0, cus = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return dmn_fail_hip(e, "gemm_nt: hipGetDevice");
    if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess || cus < 1)
        return dmn_fail_hip(e, "gemm_nt: hipDeviceGetAttribute");
    const int64_t ti = (a.M + 127) / 128;
    const int64_t slots = (int64_t)cus * nt_occupancy(NBB) / tiles_n > 0 ? (int64_t)cus * nt_occupancy(NBB) / tiles_n : 1;
    const int64_t gx = ti < slots ? ti : slots;
    hipLaunchKernelGGL(gemm_nt_kernel<NBB>, dim3((unsigned)gx, (unsigned)tiles_n), dim3(256), lds_bytes, stream, a);
    return dmn_check_launch("gemm_nt");
}

}  // namespace

#ifdef DMN_NT_TRACE
static long long* g_dmn_nt_trace = nullptr;
extern "C" int dmnerf_gemm_nt_set_trace(int64_t* d_ticks) { g_dmn_nt_trace = (long long*)d_ticks; return 0; }
#endif

extern "C" int dmnerf_gemm_nt_blocks(int n_out) {
    // out-blocks (32 outputs) per workgroup for a layer of n_out outputs: all of them up to NT_MAX_NBB, else even tiles
    const int nb = (n_out + 31) / 32;
    if (nb <= NT_MAX_NBB) return nb < 1 ? 1 : nb;
    const int tiles = (nb + NT_MAX_NBB - 1) / NT_MAX_NBB;
    return (nb + tiles - 1) / tiles;
}

extern "C" int dmnerf_pack_nt(const float* d_W, int64_t ldw, int n_rows, int c0, int k0, int c1, int k1, int transposed, const float* d_bias,
                              float* d_out, int rows_pad, int ldb, float* d_bias_out, void* stream) {
    if (n_rows < 1 || k0 < 1 || k1 < 0 || c0 < 0 || c1 < 0 || rows_pad < n_rows || rows_pad % 32)
        return dmn_fail(DMNERF_E_ARG, "pack_nt: bad sizes rows=%d k0=%d k1=%d rows_pad=%d", n_rows, k0, k1, rows_pad);
    if (ldb != ((k0 + 31) / 32 + (k1 + 31) / 32) * 32) return dmn_fail(DMNERF_E_ARG, "pack_nt: ldb=%d does not match the K ranges %d + %d", ldb, k0, k1);
    if (!d_W || !d_out) return dmn_fail(DMNERF_E_ARG, "pack_nt: null pointer");
    PackNtArgs a{d_W, ldw, n_rows, c0, k0, c1, k1, transposed, d_out, rows_pad, ldb, d_bias, d_bias_out};
    const int64_t total = (int64_t)rows_pad * ldb;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pack_nt_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream, a);
    return dmn_check_launch("pack_nt");
}

extern "C" int dmnerf_gemm_nt(const float* d_A0, int64_t lda0, int64_t a0_floats, int k0, const float* d_A1, int64_t lda1, int64_t a1_floats, int k1,
                              const float* d_B, int64_t b_floats, int ldb, const float* d_bias, float* d_C, int64_t ldc, int n_store, int n_zero,
                              int64_t M, int relu, const float* d_mask, int64_t ldm, int accumulate, void* stream) {
    if (M < 0 || k0 < 1 || k1 < 0 || n_store < 1 || n_zero < n_store) return dmn_fail(DMNERF_E_ARG, "gemm_nt: bad sizes M=%lld k0=%d k1=%d n=%d", (long long)M, k0, k1, n_store);
    if (M == 0) return DMNERF_OK;
    if (!d_A0 || !d_B || !d_C || (k1 > 0 && !d_A1)) return dmn_fail(DMNERF_E_ARG, "gemm_nt: null pointer");
    if (lda0 % 4 || (k1 > 0 && lda1 % 4) || ((uintptr_t)d_A0 & 15) || (k1 > 0 && ((uintptr_t)d_A1 & 15)) || ((uintptr_t)d_B & 15))
        return dmn_fail(DMNERF_E_ARG, "gemm_nt: operand rows must be 16-byte aligned (lda0=%lld lda1=%lld)", (long long)lda0, (long long)lda1);
    if (n_zero > ldc) return dmn_fail(DMNERF_E_ARG, "gemm_nt: n_zero=%d beyond the row length %lld", n_zero, (long long)ldc);
    const int nc0 = (k0 + 31) / 32, nc1 = (k1 + 31) / 32;
    if (ldb != 32 * (nc0 + nc1)) return dmn_fail(DMNERF_E_ARG, "gemm_nt: ldb=%d does not match the K ranges", ldb);
    if (lda0 * 4 * 128 > 0x3fffffffLL || lda1 * 4 * 128 > 0x3fffffffLL || ldc * 4 * 128 > 0x3fffffffLL || ldm * 4 * 128 > 0x3fffffffLL)
        return dmn_fail(DMNERF_E_ARG, "gemm_nt: row stride too large for 32-bit tile offsets");
    const int nbb = dmnerf_gemm_nt_blocks(n_store);
    const int tiles = ((n_store + 31) / 32 + nbb - 1) / nbb;
    if (b_floats < (int64_t)tiles * nbb * 32 * ldb) return dmn_fail(DMNERF_E_ARG, "gemm_nt: packed weights hold %lld floats, %lld needed", (long long)b_floats, (long long)tiles * nbb * 32 * ldb);
    NtArgs a{};
    a.A0 = d_A0; a.A1 = k1 > 0 ? d_A1 : nullptr; a.lda0 = lda0; a.lda1 = k1 > 0 ? lda1 : 0; a.a0_floats = a0_floats; a.a1_floats = a1_floats;
    a.nc0 = nc0; a.nc1 = nc1; a.B = d_B; a.b_floats = b_floats; a.ldb = ldb; a.bias = d_bias; a.C = d_C; a.ldc = ldc;
    a.n_store = n_store; a.n_zero = n_zero; a.mask = d_mask; a.ldm = ldm; a.M = M; a.relu = relu; a.accumulate = accumulate;
    // 16-byte epilogue accesses only where every row of C (and of the mask) keeps them 16-byte aligned
    a.x4 = (n_zero % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)d_C & 15) == 0 && (!d_mask || (ldm % 4 == 0 && ((uintptr_t)d_mask & 15) == 0))) ? 1 : 0;
#ifdef DMN_NT_TRACE
    a.trace = g_dmn_nt_trace;
#endif
    hipStream_t s = (hipStream_t)stream;
    switch (nbb) {
        case 1: return launch_nt<1>(a, tiles, s);
        case 2: return launch_nt<2>(a, tiles, s);
        case 3: return launch_nt<3>(a, tiles, s);
        case 4: return launch_nt<4>(a, tiles, s);
        case 5: return launch_nt<5>(a, tiles, s);
        case 6: return launch_nt<6>(a, tiles, s);
        case 7: return launch_nt<7>(a, tiles, s);
        case 8: return launch_nt<8>(a, tiles, s);
        case 9: return launch_nt<9>(a, tiles, s);
        case 10: return launch_nt<10>(a, tiles, s);
        default: return dmn_fail(DMNERF_E_ARG, "gemm_nt: unsupported block count %d", nbb);
    }
}

extern "C" int dmnerf_copy_cols_pad(const float* d_src, int64_t ld_src, float* d_dst, int64_t ld_dst, int64_t M, int n, int n_pad, void* stream) {
    if (M < 0 || n < 0 || n_pad < n || n_pad > ld_dst) return dmn_fail(DMNERF_E_ARG, "copy_cols_pad: bad sizes n=%d n_pad=%d ld_dst=%lld", n, n_pad, (long long)ld_dst);
    if (M == 0 || n_pad == 0) return DMNERF_OK;
    if (!d_src || !d_dst) return dmn_fail(DMNERF_E_ARG, "copy_cols_pad: null pointer");
    const int64_t total = M * n_pad;
    if ((total + 255) / 256 > 0x7fffffffLL) return dmn_fail(DMNERF_E_ARG, "copy_cols_pad: too many elements");
    hipLaunchKernelGGL(copy_cols_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_src, ld_src, d_dst, ld_dst, M, n, n_pad);
    return dmn_check_launch("copy_cols_pad");
}

extern "C" int dmnerf_ray_embed(const float* d_rays_o, const float* d_rays_d, const float* d_z, int64_t N, int S, int Lp, int Lv,
                                float* d_x_pos, int ldp, float* d_x_dir, int ldv, void* stream) {
    if (N < 0 || S < 1 || Lp < 0 || Lv < 0 || Lp > 30 || Lv > 30 || ldp < 3 + 6 * Lp || ldv < 3 + 6 * Lv)
        return dmn_fail(DMNERF_E_ARG, "ray_embed: bad sizes N=%lld S=%d Lp=%d Lv=%d ldp=%d ldv=%d", (long long)N, S, Lp, Lv, ldp, ldv);
    if (N == 0) return DMNERF_OK;
    if (!d_rays_o || !d_rays_d || !d_z || !d_x_pos || !d_x_dir) return dmn_fail(DMNERF_E_ARG, "ray_embed: null pointer");
    const int64_t blocks = (N * S + RE_ROWS - 1) / RE_ROWS;
    if (blocks > 0x7fffffffLL) return dmn_fail(DMNERF_E_ARG, "ray_embed: too many samples");
    const int lds_bytes = RE_ROWS * (ldp + 4 + ldv + 4) * 4;
    if (lds_bytes > 163840) return dmn_fail(DMNERF_E_ARG, "ray_embed: rows of %d + %d floats do not fit the staging tile", ldp, ldv);
    static DmnOncePerDevice once;
    if (hipError_t e = once.run([] { return hipFuncSetAttribute((const void*)ray_embed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); });
        e != hipSuccess)
        return dmn_fail_hip(e, "ray_embed: hipFuncSetAttribute");
    const int vec = ldp % 4 == 0 && ldv % 4 == 0 && !((uintptr_t)d_x_pos & 15) && !((uintptr_t)d_x_dir & 15);
    hipLaunchKernelGGL(ray_embed_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, (hipStream_t)stream, d_rays_o, d_rays_d, d_z, N, S, Lp, Lv,
                       d_x_pos, ldp, d_x_dir, ldv, vec);
    return dmn_check_launch("ray_embed");
}
