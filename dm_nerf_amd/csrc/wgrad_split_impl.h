// wgrad_split_impl.h -- the OPT-IN weight-gradient kernels on the 16-bit MFMA (training with args.mfma_split): one schedule, two
// operand splits (Mode): bf16x3 (wgrad_split.hip: three planes, six products) and f16x2 (wgrad_f16.hip: two planes, three).
//
// The split-K plan, the LDS ring and its LDS-DMA, the wave partition of an output tile, the partial-tile epilogue and the
// second stage are those of wgrad.hip (wgrad_common.h); only the product changes: both operands of  dW = dy . x^T  are f32
// activations, so a lane's 8 samples of a row (two ds_read_b128) are split ON THE FLY into Mode::NP planes of packed 16-bit pairs
// and one 16-sample step of a tile pair is Mode::NT 32x32x16 MFMAs instead of eight v_mfma_f32_32x32x2_f32: 16 NT NBA NBB MFMA
// cycles per 32-sample chunk instead of 256 NBA NBB.
//
// Schedule: a chunk is 2 steps of 16 samples; a step is IS "items" (GA A blocks against the wave's SBn B blocks = NT GA SBn
// MFMAs, at least four accumulators in rotation wherever the tile has them).  While the MFMAs of item i run, the raw A
// operands of item i + 1 are split and the LDS reads of item i + 2 are issued; the B operands of the NEXT step are read in a
// step's first item (the hand-over item when that step belongs to the next chunk) and split over the rest of the step.  All
// of it is placed per slot: an item has NG slots = its MFMAs (or a few more where the tile is too small to give every read its
// own slot: only with NT = 3), the split of an operand pair is cut into two units (Mode::unit) and the units of a chunk are
// spread evenly (no packed f32 VALU: the TUs are built with -fno-slp-vectorize).  B planes are double-buffered by step, A planes
// and raw operands by item.  The ring hand-over sits at the start of the chunk's last-but-one item (every read of the chunk has
// returned by then), the refill pieces follow it three slots apart.
// f32-class results (not the bitwise fmaf chain of wgrad.hip), hence opt-in.
#pragma once
#include "wgrad_common.h"

extern long long* g_dmn_wgrad_trace;      // wgrad.hip

namespace {

// Wave partition of an output tile for this kernel: a 2 x 2 wave grid wherever the tile allows it -- a wave then splits
// SAn + SBn = (NBA + NBB) / 2 operand blocks per step instead of NBA + NBB / 4 (every split is VALU work that the MFMAs of a
// one-wave-per-SIMD kernel do not hide) and owns SBn >= 2 accumulators per A block.  Same interface as wgrad_common.h::Split.
template <int NBA, int NBB>
struct SplitS {
    static constexpr bool G22 = (NBB >= 4 && NBA % 2 == 0) || NBB == 2;
    static constexpr int WA = G22 ? 2 : (NBB == 1 ? 4 : 1), WB = 4 / WA;          // wave grid
    static_assert(NBA % WA == 0 && NBB % WB == 0, "unsupported shape class");
    static constexpr int SAn = NBA / WA, SBn = NBB / WB;
    static constexpr int NSHARE = WB;                                             // waves that hold the same A blocks
    static constexpr int HOST_NSHARE = NBB >= 4 ? 4 : (NBB == 2 ? 2 : 1);         // shares the plan reserves (Split<>::NSHARE)
    static_assert(HOST_NSHARE % NSHARE == 0, "row-sum shares");
    __device__ static constexpr int a_lit(int k) { return WA * k; }
    __device__ static int a_wave(int w) { return WA == 1 ? 0 : (WA == 2 ? (w >> 1) : w); }
    __device__ static constexpr int b_lit(int k) { return WB * k; }
    __device__ static int b_wave(int w) { return WB == 1 ? 0 : (WB == 2 ? (w & 1) : w); }
    __device__ static int share_rank(int w) { return b_wave(w); }
};

// Work beside the MFMAs is placed per MFMA gap: a one-wave-per-SIMD kernel hides ~5 single-issue instructions per 16-bit
// 32x32x16 MFMA (MI355X_MICROARCH.md), so the split of an operand pair is cut into two units (Mode::unit) and the units of a chunk
// are spread evenly.
constexpr int spread(int u, int n, int g0, int g1) { return g0 + (int)((long long)u * (g1 - g0) / n); }      // unit u of n over gaps [g0, g1)
#ifndef DMN_WGS_REFILL_SPREAD
#define DMN_WGS_REFILL_SPREAD 3            // gaps per refill piece after the hand-over
#endif
constexpr int refill_gap(int p, int NG2, int NL) {
    const int g = p * DMN_WGS_REFILL_SPREAD;
    return (NL - 1) * DMN_WGS_REFILL_SPREAD < NG2 ? g : p * NG2 / NL;
}

template <class Mode, int NBA, int NBB>
__device__ __forceinline__ void run_job_split(const WgArgs& a, const WgJob& jb, float* lds) {
    constexpr int NT = Mode::NT, NP = Mode::NP, NTMP = Mode::NTMP;
    typedef SplitS<NBA, NBB> SP;
    typedef Ring<NBA, NBB> RG;
    constexpr int SAn = SP::SAn, SBn = SP::SBn, NPAIR = SAn * SBn;
    // an item = GA A blocks x the wave's SBn B blocks (>= 4 accumulators in rotation wherever the tile has them: back-to-back
    // MFMAs on one accumulator wait for each other).  A full step per item except for the 16-accumulator tile, whose planes
    // and raw operands of a whole step do not fit beside the 256 accumulator registers.
    constexpr int GA = NPAIR >= 16 ? SAn / 4 : SAn;
    constexpr int NACC = GA * SBn;
    constexpr int IS = SAn / GA;                   // items per 16-sample step (1 or 4)
    constexpr int NI = 2 * IS;                     // items per chunk
    constexpr int NGM = NT * NACC;                 // MFMAs per item
    constexpr int NUA = 8 * GA, NUB = 8 * SBn, NUS = 2 * GA;      // split units of an item's A blocks / of a step's B blocks; row-sum units
    constexpr int NRA = 2 * GA, NRB = 2 * SBn;     // LDS reads of an item's A blocks / of a step's B blocks
    constexpr int RB = IS == 1 ? 2 : 1;            // raw B buffers
    constexpr int NL = NBA + NBB;                  // DMA pieces per wave per chunk (1 KiB each)
    constexpr int H = NI - 2;                      // the hand-over sits at the start of this item
    constexpr int NG = NGM >= NRA + NRB && 2 * NGM >= NL ? NGM : (NRA + NRB > (NL + 1) / 2 ? NRA + NRB : (NL + 1) / 2);   // slots per item
    constexpr int WB = NRA + NRB + Mode::B_WAIT;   // IS > 1: slot (counted from the item that issues a step's B reads) from which its B units may run
    constexpr int D = RG::D, BUF = RG::BUF;
    static_assert(IS == 1 || IS == 4, "items per step");
    static_assert((D - 1) * NL <= 63, "vmcnt range");
    static_assert(NG >= NRA + NRB && 2 * NG >= NL && (IS == 1 || WB < 2 * NG), "slots for the reads / the refill");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));               // opaque per item: the lane geometry below is recomputed for each item of a workgroup
                                               // (hoisted out of the item loop, nine shape classes' worth of it would spill)
    const int lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = lds_addr(lds);

    // ---- DMA geometry (as wgrad.hip): wave w owns the 1-KiB pieces q = w + 4 j of a tile; 16-byte units XOR-swizzled by row
    const int drow = 8 * w + (lane >> 3);
    const int dvoff = drow * 128 + (((lane & 7) ^ ((drow >> 1) & 7)) << 4);
    const float* __restrict__ A = a.src[jb.a_src] + jb.a_off + (int64_t)jb.a_row0 * 32;
    const float* __restrict__ B = a.src[jb.b_src] + jb.b_off + (int64_t)jb.b_row0 * 32;
    const int64_t strideA = (int64_t)jb.a_R * 32, strideB = (int64_t)jb.b_R * 32;
    const int nchunk = jb.nchunk;
    // the two tile descriptors of chunk c (clamped: the ring's last refills re-fetch the last chunk) -- built once per chunk
    auto chunk_rsrc = [&](int c, rsrc_t& rsA, rsrc_t& rsB) {
        const int cc = c < nchunk ? c : nchunk - 1;
        rsA = uniform_rsrc(A + (int64_t)(jb.chunk0 + cc) * strideA, (int64_t)jb.rowsA * 32);
        rsB = uniform_rsrc(B + (int64_t)(jb.chunk0 + cc) * strideB, (int64_t)jb.rowsB * 32);
    };
    auto dma_piece = [&](const rsrc_t& rsA, const rsrc_t& rsB, unsigned slot_byte, int i) {
        const bool isA = i < NBA;
        const int j = isA ? i : i - NBA;
        float* dst = lds + (slot_byte + i * 4096 + w * 1024) / 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rsA : rsB, (DMN_LAS void*)dst, 16, dvoff, j * 4096, 0, 0);
    };

    // ---- read geometry: lane (li, half) reads row 32 blk + li, units 4 s + half and 4 s + 2 + half in step s: its 8 k-slots
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

    f32x4 rawA[2][GA][2], rawB[RB][SBn][2];           // [item parity][block][read], [step parity if RB == 2][block][read]
    unsigned PA[2][GA][NP][4], PB[2][SBn][NP][4];       // [item parity][block][plane][word], [step parity][block][plane][word]
    float tA[GA][4][NTMP], tB[SBn][4][NTMP];            // what a pair's first unit leaves for its second (residuals)

    // unit u = 2 q + stage of the block (r -> P): Mode::unit
    auto unit = [](const f32x4 (&r)[2], unsigned (&P)[NP][4], float (&t)[4][NTMP], auto uc) { Mode::template unit<decltype(uc)::value>(r, P, t); };
    auto read_a = [&](auto itc, auto gc, unsigned slot) {           // read g of the A blocks of item `it` (ring slot at byte `slot`)
        constexpr int it = decltype(itc)::value, g = decltype(gc)::value;
        constexpr int s = it / IS, ig = it % IS;
        lds_read16_async<SP::a_lit(ig * GA + (g >> 1)) * 4096>(rawA[it & 1][g >> 1][g & 1], offA[2 * s + (g & 1)] + slot);
    };
    auto read_b = [&](auto sc, auto gc, unsigned slot) {            // read g of the B blocks of step s
        constexpr int s = decltype(sc)::value, g = decltype(gc)::value;
        lds_read16_async<SP::b_lit(g >> 1) * 4096>(rawB[s & (RB - 1)][g >> 1][g & 1], offB[2 * s + (g & 1)] + slot);
    };
    auto wait_a = [&](auto itc) {
        constexpr int it = decltype(itc)::value;
#pragma unroll
        for (int k = 0; k < GA; ++k) lds_wait<0>(rawA[it & 1][k]);
    };
    auto wait_b = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int ib = 0; ib < SBn; ++ib) lds_wait<0>(rawB[s & (RB - 1)][ib]);
    };
    // row sums of dy ride on the raw A operands; the steps are dealt out over the NSHARE waves that hold the same A blocks:
    // unit v = 2 k + read of item `it`, 4 v_fma_f32 with the wave-uniform factor 1 / 0
    auto bias_unit = [&](auto itc, auto vc, float f) {
        constexpr int it = decltype(itc)::value, v = decltype(vc)::value;
        constexpr int ig = it % IS, k = v >> 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[ig * GA + k][e] = __builtin_fmaf(rawA[it & 1][k][v & 1][e], f, bs[ig * GA + k][e]);
        asm volatile("" : "+v"(bs[ig * GA + k]));
    };
    auto bias_factor = [&](int step_global, int chunk) {
        return (want_bias && (step_global & (SP::NSHARE - 1)) == my_rank && chunk < nchunk) ? 1.f : 0.f;
    };

    // ---- prologue: D chunks in flight, chunk 0 landed; item 0 split, item 1 on its way
#pragma unroll
    for (int sl = 0; sl < D; ++sl) {
        rsrc_t rsA, rsB;
        chunk_rsrc(sl, rsA, rsB);
#pragma unroll
        for (int i = 0; i < NL; ++i) dma_piece(rsA, rsB, sl * BUF, i);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 1) * NL) & 15) | ((((D - 1) * NL) >> 4) << 14));     // vmcnt((D-1) NL) only
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        typedef std::integral_constant<int, 0> Z;
        typedef std::integral_constant<int, 1> One;
        static_for<NRA>([&](auto gc) { read_a(Z{}, gc, 0u); });
        static_for<NRB>([&](auto gc) { read_b(Z{}, gc, 0u); });
        wait_a(Z{});
        wait_b(Z{});
#pragma unroll
        for (int k = 0; k < GA; ++k) static_for<8>([&](auto uc) { unit(rawA[0][k], PA[0][k], tA[k], uc); });
#pragma unroll
        for (int ib = 0; ib < SBn; ++ib) static_for<8>([&](auto uc) { unit(rawB[0][ib], PB[0][ib], tB[ib], uc); });
        const float f0 = bias_factor(0, 0);
        static_for<NUS>([&](auto vc) { bias_unit(Z{}, vc, f0); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<NRA>([&](auto gc) { read_a(One{}, gc, 0u); });
        if constexpr (IS == 1) static_for<NRB>([&](auto gc) { read_b(One{}, gc, 0u); });
    }

    unsigned sb = 0;                                    // byte offset of the ring slot of chunk c (uniform)
#pragma nounroll
    for (int c = 0; c < nchunk; ++c) {
        const unsigned nb = sb + BUF == (unsigned)(D * BUF) ? 0u : sb + BUF;
        rsrc_t rfA, rfB;                                                  // the refill (chunk c + D)
        chunk_rsrc(c + D, rfA, rfB);
        static_for<NI>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int s = i / IS, ig = i % IS;
            constexpr int i1 = (i + 1) % NI, n1 = (i + 1) / NI;          // the item whose A operands are split now
            constexpr int s1 = i1 / IS, ig1 = i1 % IS;
            constexpr int i2 = (i + 2) % NI, n2 = (i + 2) / NI;          // the item whose A reads are issued now
            constexpr int s2 = i2 / IS;
            constexpr int sn = (s + 1) & 1;                              // the next step (B operands)
            // IS > 1: the B operands of the next step are read in the step's first item -- the hand-over item when the step
            // belongs to the next chunk -- and split from gap WB of that item to the end of the step
            constexpr int ibr = s == 0 ? 0 : H;                          // (IS > 1) item of step s that issues them
            constexpr int bspan = (NI / 2 - (ibr - s * IS)) * NG - WB;   // slots their units are spread over
            static_assert(IS == 1 || bspan >= 4, "B units");
            wait_a(std::integral_constant<int, i1>{});
            if constexpr (IS == 1) wait_b(std::integral_constant<int, s1>{});
            if constexpr (i == H) {
                // ring hand-over: chunk c + 1 has landed in every wave's view, and this chunk's slot is released (every read
                // of it was issued at least one item ago and has returned: lgkmcnt(0) above)
                __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 2) * NL) & 15) | ((((D - 2) * NL) >> 4) << 14));
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            const float fb = bias_factor(2 * (c + n1) + s1, c + n1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int term = g / NACC, pr = g % NACC, ka = pr / SBn, ib = pr % SBn;
                // -- LDS reads: the A blocks of item i + 2; the B blocks of the next step
                if constexpr (g < NRA) read_a(std::integral_constant<int, i2>{}, gc, n2 ? nb : sb);
                if constexpr (IS == 1) {
                    if constexpr (g >= NRA && g < NRA + NRB) read_b(std::integral_constant<int, s2>{}, std::integral_constant<int, g - NRA>{}, n2 ? nb : sb);
                } else {
                    if constexpr (i == ibr && g >= NRA && g < NRA + NRB) read_b(std::integral_constant<int, sn>{}, std::integral_constant<int, g - NRA>{}, s == 1 ? nb : sb);
                    if constexpr (i >= ibr && (i - ibr) * NG + g == WB) wait_b(std::integral_constant<int, sn>{});      // (issued >= Mode::B_WAIT slots ago)
                }
                // -- split units of the A blocks of item i + 1
                static_for<NUA>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    if constexpr (spread(u, NUA, 0, NG) == g) unit(rawA[i1 & 1][u >> 3], PA[i1 & 1][u >> 3], tA[u >> 3], std::integral_constant<int, (u & 7)>{});
                });
                // -- split units of the B blocks of the next step
                static_for<NUB>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    constexpr bool here = IS == 1 ? (ig1 == 0 && spread(u, NUB, NG / (2 * NUB), NG) == g)
                                                  : (i >= ibr && spread(u, NUB, WB + 1, WB + 1 + bspan - 1) == (i - ibr) * NG + g);
                    if constexpr (here) unit(rawB[sn & (RB - 1)][u >> 3], PB[sn][u >> 3], tB[u >> 3], std::integral_constant<int, (u & 7)>{});
                });
                // -- row sums (the raw A operands of item i + 1)
                static_for<NUS>([&](auto vc) {
                    constexpr int v = decltype(vc)::value;
                    if constexpr (spread(v, NUS, NG / 2, NG) == g) bias_unit(std::integral_constant<int, i1>{}, vc, fb);
                });
                if constexpr (i >= H)                                     // refill the released slot with chunk c + D
                    static_for<NL>([&](auto pc) {
                        constexpr int p = decltype(pc)::value;
#ifdef DMN_WGS_NODMA           // timing experiment: no refills (stale operands)
                        if constexpr (false)
#endif
                        if constexpr (refill_gap(p, 2 * NG, NL) == (i - H) * NG + g) dma_piece(rfA, rfB, sb, p);
                    });
                constexpr int ai = (ig * GA + ka) * SBn + ib;
#ifdef DMN_WGS_NOMFMA          // timing experiment: everything but the MFMAs (one in six kept for the dependences)
                if constexpr (term == 0)
#endif
                if constexpr (g < NGM)
                    {
                    constexpr int ta = Mode::term_a(term < NT ? term : 0), tb = Mode::term_b(term < NT ? term : 0);
                    acc[ai] = Mode::mfma(PA[i & 1][ka][ta], PB[s & 1][ib][tb], acc[ai]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        sb = nb;
    }
    // the ring's last (clamped) refills and the read-ahead past the last chunk: both land in registers / LDS nobody uses,
    // but they must have landed before the epilogue reuses either
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < GA; ++j) lds_wait<0>(rawA[k][j]);
#pragma unroll
    for (int k = 0; k < RB; ++k)
#pragma unroll
        for (int ib = 0; ib < SBn; ++ib) lds_wait<0>(rawB[k][ib]);
    store_partials<SP, NBA, NBB>(a, jb, acc, bs, w, half, li, want_bias, my_rank);
}

// The kernel body; the translation units wrap it in their own __global__ function (wgrad_split_kernel / wgrad_f16_kernel: the
// names the profiles and bench.py's roofline entries carry).
template <class Mode>
__device__ __forceinline__ void wgrad_split_body(const WgArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    WgJob jb = a.jobs[blockIdx.x];
    if (a.trace && threadIdx.x == 0) a.trace[2 * blockIdx.x] = (long long)wall_clock64();
    for (int item = jb.next - 1, more = jb.follow;; --more) {      // (block = leader item; its followers start at jb.next)
        switch (jb.cls) {                              // workgroup-uniform
            case C_8_8: run_job_split<Mode, 8, 8>(a, jb, lds); break;
            case C_4_8: run_job_split<Mode, 4, 8>(a, jb, lds); break;
            case C_8_2: run_job_split<Mode, 8, 2>(a, jb, lds); break;
            case C_4_1: run_job_split<Mode, 4, 1>(a, jb, lds); break;
            case C_1_8: run_job_split<Mode, 1, 8>(a, jb, lds); break;
            case C_1_4: run_job_split<Mode, 1, 4>(a, jb, lds); break;
            case C_2_4: run_job_split<Mode, 2, 4>(a, jb, lds); break;
            case C_3_4: run_job_split<Mode, 3, 4>(a, jb, lds); break;
            case C_4_4: run_job_split<Mode, 4, 4>(a, jb, lds); break;
            default: break;
        }
        if (more <= 0) break;
        __syncthreads();                               // every wave is done with the LDS ring: the next item refills it
        jb = a.jobs[++item];
    }
    if (a.trace && threadIdx.x == 0) a.trace[2 * blockIdx.x + 1] = (long long)wall_clock64();
}

// Launch + second stage; d_unscale: null, or the factor the dy operands must lose again (f16x2: 2^-s of dmnerf_grad_scale).
template <class Kernel>
int wgrad_split_launch(Kernel kernel, DmnOncePerDevice& once, const char* what, const float* d_save, const float* d_dsave, const float* d_graw_t, int64_t M, const void* d_jobs, int n_jobs,
                       const void* d_outs, int n_outs, const float* d_params_flat, int ins_num, float* d_part, float* d_grad_flat,
                       const float* d_unscale, void* stream) {
    if (!d_save || !d_dsave || !d_graw_t || !d_jobs || !d_outs || !d_params_flat || !d_part || !d_grad_flat || M < 1 || n_jobs < 1 || n_outs < 1)
        return dmn_fail(DMNERF_E_ARG, "%s: bad argument", what);
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "%s: ins_num %d unsupported", what, ins_num);
    WgArgs a{};
    a.src[0] = d_save; a.src[1] = d_dsave; a.src[2] = d_graw_t;
    a.part = d_part; a.jobs = (const WgJob*)d_jobs; a.Mp = save_row_len(M); a.trace = g_dmn_wgrad_trace;
    const size_t lds_bytes = WG_LDS_BYTES;
    if (hipError_t e = once.run([&] { return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); });
        e != hipSuccess)
        return dmn_fail_hip(e, what);
    hipLaunchKernelGGL(kernel, dim3((unsigned)n_jobs), dim3(256), lds_bytes, (hipStream_t)stream, a);
    const int rc = dmn_check_launch(what);
    if (rc) return rc;
    return dmn_wgrad_finish(d_outs, n_outs, d_params_flat, ins_num, d_part, d_grad_flat, (hipStream_t)stream, d_unscale);
}

}  // namespace

