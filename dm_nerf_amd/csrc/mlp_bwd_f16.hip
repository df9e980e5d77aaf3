// mlp_bwd_f16.hip -- OPT-IN data-gradient kernel on the split-f16 MFMA path (args.mfma_split = "f16x2" in training).
//
// The dgrad of mlp_bwd.hip (ins_linear^T | F^T | mlps.7^T .. mlps.1^T, re-associated heads, layout.h::BlobTLayout) with the
// engine of the split-f16 forward (mlp_f16_impl.h): W^T pre-split into two f16 planes (layout.h::F16TLayout), the gradient a
// layer consumes kept as two planes of packed f16 pairs, three MFMAs per f32 product accumulated in f32, the network walked
// OUT-BLOCK-OUTER -- a pass accumulates two blocks of dh_{l-1} over all 16 k-blocks of dy_l -- so that the epilogue of a pass
// (ReLU mask from the forward's bit words, f32 row store for the weight-gradient kernel, plane split) rides in the MFMA gaps
// of the next one.  What it reads (bit masks, dL/draw) and writes (dy rows in the f32 SaveLayout workspace, d raw transposed)
// is exactly what mlp_bwd.hip reads and writes.  f32-class, not the bitwise f32 chain of the default kernel: opt-in.
#include "mlp_f16_impl.h"

namespace {

struct BwdHArgs {
    const float* blob;     // [table TAB_T_FLOATS f32 | W^T group stream]
    BlobLayout L;
    BlobTLayout LT;        // table offsets (w_rgbo, w_den)
    F16TLayout S;
    const float* save;     // forward workspace (masks)
    const float* graw;     // [M, 4+C]
    float* dsave;          // gradients, same SaveLayout
    float* graw_t;         // d raw block-major [blk][4+C][32]
    const float* scale;    // nullable: {2^s, 2^-s} from dmnerf_grad_scale -- dL/draw is multiplied by 2^s on the way in
    int64_t M;
};

// f16 has 5 exponent bits: the data gradients of a real training step (dL/draw ~ 1e-4 / rays, dy smaller still) would sit in
// its subnormal range.  The whole backward is LINEAR in dL/draw, so it runs on 2^s dL/draw with s = 6 - ceil(log2 max|dL/draw|)
// (max 64: three decades of headroom for growth through the layers, 2^-2 .. 64 at full two-plane precision) and the weight-
// gradient reduction multiplies by 2^-s -- both exact.  One pass over dL/draw; the last block to finish writes the factors.
__global__ __launch_bounds__(256) void grad_scale_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out4) {
    __shared__ unsigned wmax[4];
    unsigned mx = 0u;
    auto take = [&](float x) {
        const unsigned u = __float_as_uint(x) & 0x7fffffffu;
        mx = (u < 0x7f800000u && u > mx) ? u : mx;               // (inf / nan do not set the scale)
    };
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n / 4 : 0;           // 16-byte lanes where the buffer allows them
    const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(g);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {               // four loads in flight per lane
        const f32x4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
#pragma unroll
        for (int e = 0; e < 4; ++e) { take(v0[e]); take(v1[e]); take(v2[e]); take(v3[e]); }
    }
    for (; i < n4; i += stride) {
        const f32x4 v = g4[i];
        take(v[0]); take(v[1]); take(v[2]); take(v[3]);
    }
    for (int64_t j = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) take(g[j]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned v = (unsigned)__shfl_xor((int)mx, o); mx = v > mx ? v : mx; }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    unsigned* scratch = reinterpret_cast<unsigned*>(out4) + 2;    // [2] = running max (bits), [3] = blocks done; both zero between calls
    if (threadIdx.x == 0) {                                       // ONE atomic pair per block (a grid of a few hundred: the atomics were
        const unsigned a01 = wmax[0] > wmax[1] ? wmax[0] : wmax[1], a23 = wmax[2] > wmax[3] ? wmax[2] : wmax[3];   // the kernel's time at 2048 x 4)
        atomicMax(scratch, a01 > a23 ? a01 : a23);
        __threadfence();
        if (atomicAdd(scratch + 1, 1u) == gridDim.x - 1) {
            const unsigned all = atomicMax(scratch, 0u);
            float sc = 1.f;
            if (all != 0u) {
                int e;
                (void)frexpf(__uint_as_float(all), &e);          // max = f 2^e, f in [0.5, 1)  ->  2^(6 - e) max in [32, 64)
                int s = 6 - e;
                s = s > 120 ? 120 : (s < -120 ? -120 : s);
                sc = ldexpf(1.f, s);
            }
            out4[0] = sc;
            out4[1] = 1.f / sc;
            scratch[0] = 0u;
            scratch[1] = 0u;
            __threadfence();
        }
    }
}


// The epilogue of a finished dgrad pass, dealt out over the MFMA gaps of the next one like mlp_f16_impl.h::EpiFwd:
//   phase 0 : 2 v_accvgpr_read (+ HASQ: the density term  w_d g_sigma, one fma each)
//   phase 1 : ReLU mask from the forward's bit word (element p = 32 w + i is bit 31 - i): v_bfe_i32 + v_and per element
//   phase 2 : the two TID-addressed f32 row stores (what wgrad.hip reads) + SPLIT: the hi word
//   phase 3 : SPLIT: the 2 residuals + the lo word
// NUM pairs per burst of NPH gaps; MW = the 32-bit mask words of the pass's NOBP blocks (one per two blocks).
template <int NOBP, int OB0, int QB0, bool HASQ, bool SPLIT, int G0, int NUM, int NWO, int NBQA>
struct EpiBwd {
    static constexpr int NPH = SPLIT ? 4 : 3;
    f32x16 (&Y)[NOBP];
    f32x4 (&bq)[NBQA];
    unsigned (&Ohi)[NWO];
    unsigned (&Olo)[NWO];
    RowIO io;
    unsigned mword[(NOBP + 1) / 2];
    float gs;                                   // HASQ: g_sigma
    float xs0[NUM] = {}, xs1[NUM] = {};
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
            constexpr int w = SPLIT ? (2 * (OB0 + b) + (r >> 3)) * 4 + ((r & 7) >> 1) : 0;
            static_assert(w < NWO, "plane word");
            if constexpr (ph == 0) {
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(Y[b][r]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(Y[b][r + 1]));
                if constexpr (HASQ) {
                    x0 = fmaf(bq[QB0 + b * 4 + (r >> 2)][r & 3], gs, x0);
                    x1 = fmaf(bq[QB0 + b * 4 + ((r + 1) >> 2)][(r + 1) & 3], gs, x1);
                }
                asm volatile("" : "+v"(x0), "+v"(x1));
            } else if constexpr (ph == 1) {
                constexpr int p0 = 16 * b + r;                   // element index inside the pass: word p0 >> 5, bit 31 - (p0 & 31)
                const unsigned k0 = (unsigned)__builtin_amdgcn_sbfe((int)mword[p0 >> 5], 31 - (p0 & 31), 1);
                const unsigned k1 = (unsigned)__builtin_amdgcn_sbfe((int)mword[p0 >> 5], 30 - (p0 & 31), 1);
                x0 = __uint_as_float(__float_as_uint(x0) & k0);
                x1 = __uint_as_float(__float_as_uint(x1) & k1);
                asm volatile("" : "+v"(x0), "+v"(x1));
            } else if constexpr (ph == 2) {
                DMN_ACT_STORE_B32(f2u(x0), io.rs, run_off(0, r), (int)(io.soff + (OB0 + b) * 4096), DMN_STORE_AUX);
                DMN_ACT_STORE_B32(f2u(x1), io.rs, run_off(0, r + 1), (int)(io.soff + (OB0 + b) * 4096), DMN_STORE_AUX);
                if constexpr (SPLIT) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(Ohi[w]) : "v"(x0), "v"(x1));
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

// two epilogues riding in the same pass
template <class A_, class B_>
struct Both {
    A_& a;
    B_& b;
    template <int GAP>
    __device__ __forceinline__ void operator()(std::integral_constant<int, GAP> g) { a(g); b(g); }
};

template <int OBI>
__global__ __launch_bounds__(256) void mlp_bwd_f16_kernel(const BwdHArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [ring 8 x 16 KiB][table 4 KiB]
    float* const tab = lds + F16_RING_FLOATS;
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
    const float gsc = a.scale ? a.scale[0] : 1.f;                // power of two: every product below is exact
    const float g_rgb[3] = {valid ? gr[0] * gsc : 0.f, valid ? gr[1] * gsc : 0.f, valid ? gr[2] * gsc : 0.f};
    const float g_sigma = valid ? gr[3] * gsc : 0.f;
    const int GR = 4 + L.C;
    rsrc_t grs = uniform_rsrc(a.graw_t, a.graw_t ? (int64_t)GR * MP : 0);
    const int gv = (int)((blk * GR * 32 + (lane & 31)) * 4);
    unsigned Pgi[2][OBI * 8];
    {
        f32x16 gi[OBI];
#pragma unroll
        for (int b = 0; b < OBI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
                const bool in = ch < L.C;
                const float v = gr[4 + (in ? ch : L.C - 1)];
                gi[b][r] = (in && valid) ? v * gsc : 0.f;
                __builtin_amdgcn_raw_buffer_store_b32(f2u(gi[b][r]), grs, in ? gv + (4 + ch) * 128 : 0x7ffffff0, 0, 0);
            }
        split_blocks_f16<OBI>(gi, Pgi[0], Pgi[1]);
    }
    if (half == 0) {
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_rgb[0]), grs, gv, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_rgb[1]), grs, gv + 128, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_rgb[2]), grs, gv + 256, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(f2u(g_sigma), grs, gv + 384, 0, 0);
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
    reinterpret_cast<f32x4*>(tab)[threadIdx.x] = (reinterpret_cast<const f32x4*>(a.blob) + threadIdx.x)[0];   // TAB_T_FLOATS = 256 x float4

    GStream ws;
    ws.rs = uniform_rsrc(a.blob, a.S.total);
    ws.wave = wave;
    ws.voff = (unsigned)(lane * 16 + wave * 1024);
    ws.ring = lds;
    ws.off = __builtin_amdgcn_readfirstlane((unsigned)(a.S.stream * 4));
    ws.gidx = 0;
    ws.lane16 = lds_addr(lds) + lane * 16;
#pragma unroll
    for (int g = 0; g < F16_LA; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gs_fetch_piece(ws, g, i);
        ws.off += F16_GROUP_BYTES;
    }
    // everything but the youngest 16 pieces (groups 2 .. 5) has landed / returned: groups 0 and 1, the masks, the table source
    wait_vm<4 * (F16_LA - 2)>();
    wait_lgkm<0>();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- dg1 = relu'(g1) . (W_rgbo^T g_rgb) on the VALU (3 terms per element): rows for the weight gradient + planes
    unsigned Pg1[2][32];
    {
        const RowIO gio = make_rowio(a.dsave + SL.g1, 256, MP, blk, lane);              // (dg1 | dg2: one 256-row tensor, layout.h)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float x[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int p = 16 * b + r + e;
                    float t = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) t = fmaf(tab[LT.w_rgbo + (c * 2 + half) * 64 + p], g_rgb[c], t);
                    const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)g1bits[p >> 5], 31 - (p & 31), 1);
                    x[e] = __uint_as_float(__float_as_uint(t) & keep);
                    DMN_ACT_STORE_B32(f2u(x[e]), gio.rs, run_off(0, r + e), (int)(gio.soff + b * 4096), DMN_STORE_AUX);
                }
                const int w = (2 * b + (r >> 3)) * 4 + ((r & 7) >> 1);
                split_pair_f16(x[0], x[1], Pg1[0][w], Pg1[1][w]);
            }
    }
    // (the compiler's own LDS reads of the table above are behind its own waits; the asm reads start here)
    wait_lgkm<0>();
    ws.cur = ws.lane16;
    ws.nxt = ws.lane16 + F16_GROUP_BYTES;
    static_for<8>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_read16_async<i * 1024>(ws.H[i], ws.cur); });

    f32x16 acc0[2], acc1[2], accG[4];
    unsigned PA[2][64], PB[2][64];
    f32x4 bq[16];
    // density term of dh_7: this lane's w_den quads, 16 floats per block
    unsigned baddr = lds_addr(tab) + (unsigned)(LT.w_den + half * 128) * 4;
    constexpr int VMT = 16;                     // row stores guaranteed younger than the awaited pieces at a trunk hand-over (see mlp_f16_impl.h)

    // ---- dg2 = relu'(g2) . (W_io^T g_ins)   (nothing flows on to h_7: the ins branch starts from h.detach(), dm_nerf.py:95)
    f16_pass<4, OBI, 0, 0, 0, 0, true, 0>(ws, Pgi[0], Pgi[1], accG, bq, baddr, NoSideC{});
    // ---- dh_7 = F^T dg1 + w_d g_sigma; dy_7 = dh_7 . relu'(h_7): four passes of two groups; the first also carries dg2's epilogue
    EpiBwd<4, 0, 0, false, false, 3, 4, 64, 16> eG{accG, bq, PA[0], PA[1], make_rowio(a.dsave + SL.g1, 256, MP, blk, lane, 4), {g2bits[0], g2bits[1]}, 0.f};
    const RowIO io7 = make_rowio(a.dsave + SL.h + (int64_t)7 * 256 * MP, 256, MP, blk, lane);
    f16_pass<2, 2, 0, 8, 0, 0, true, 0, 64>(ws, Pg1[0], Pg1[1], acc0, bq, baddr, eG);
    baddr += 128;
    f16_pass<2, 2, 0, 8, 8, 0, true, 0, 64>(ws, Pg1[0], Pg1[1], acc1, bq, baddr, EpiBwd<2, 0, 0, true, true, 3, 2, 64, 16>{acc0, bq, PA[0], PA[1], io7, {hbits[7][0]}, g_sigma});
    baddr += 128;
    f16_pass<2, 2, 0, 8, 0, 0, true, 0, 64>(ws, Pg1[0], Pg1[1], acc0, bq, baddr, EpiBwd<2, 2, 8, true, true, 3, 2, 64, 16>{acc1, bq, PA[0], PA[1], io7, {hbits[7][1]}, g_sigma});
    baddr += 128;
    f16_pass<2, 2, 0, 8, 8, 0, true, 0, 64>(ws, Pg1[0], Pg1[1], acc1, bq, baddr, EpiBwd<2, 4, 0, true, true, 3, 2, 64, 16>{acc0, bq, PA[0], PA[1], io7, {hbits[7][2]}, g_sigma});

    // ---- trunk: dh_{l-1} = W_l^T dy_l, l = 7 .. 1; dy_{l-1} = dh_{l-1} . relu'(h_{l-1}) -> rows (+ planes unless l = 1).
    // Pass p accumulates out-blocks 2p, 2p+1 in set p & 1 while the other set is post-processed; the first pass of a layer
    // post-processes the previous layer's last pass into the k-blocks 12..15 it needs in its fourth group.
    auto layer = [&](auto lc, unsigned (&Pin)[2][64], unsigned (&Pout)[2][64], auto first) __attribute__((always_inline)) {
        constexpr int LAYER = decltype(lc)::value;                       // W_LAYER^T: consumes dy_LAYER, produces dy_{LAYER-1}
        constexpr bool FIRST = decltype(first)::value;                   // (mlps.7^T: the pending pass is dh_7's last, with the density term)
        constexpr bool SP = LAYER > 1;                                   // dy_0 has no consumer
        const RowIO iop = make_rowio(a.dsave + SL.h + (int64_t)LAYER * 256 * MP, 256, MP, blk, lane);
        const RowIO iol = make_rowio(a.dsave + SL.h + (int64_t)(LAYER - 1) * 256 * MP, 256, MP, blk, lane);
        f16_pass<2, 4, 0, 0, 0, 0, true, VMT>(ws, Pin[0], Pin[1], acc0, bq, baddr,
                                              EpiBwd<2, 6, 8, FIRST, true, 3, 1, 64, 16>{acc1, bq, Pin[0], Pin[1], iop, {hbits[LAYER][3]}, g_sigma});
        f16_pass<2, 4, 0, 0, 0, 0, true, VMT>(ws, Pin[0], Pin[1], acc1, bq, baddr,
                                              EpiBwd<2, 0, 0, false, SP, 3, 1, 64, 16>{acc0, bq, Pout[0], Pout[1], iol, {hbits[LAYER - 1][0]}, 0.f});
        f16_pass<2, 4, 0, 0, 0, 0, true, VMT>(ws, Pin[0], Pin[1], acc0, bq, baddr,
                                              EpiBwd<2, 2, 0, false, SP, 3, 1, 64, 16>{acc1, bq, Pout[0], Pout[1], iol, {hbits[LAYER - 1][1]}, 0.f});
        f16_pass<2, 4, 0, 0, 0, 0, true, VMT>(ws, Pin[0], Pin[1], acc1, bq, baddr,
                                              EpiBwd<2, 4, 0, false, SP, 3, 1, 64, 16>{acc0, bq, Pout[0], Pout[1], iol, {hbits[LAYER - 1][2]}, 0.f});
    };
    typedef std::true_type T_;
    typedef std::false_type F_;
    layer(std::integral_constant<int, 7>{}, PA, PB, T_{});
    layer(std::integral_constant<int, 6>{}, PB, PA, F_{});
    layer(std::integral_constant<int, 5>{}, PA, PB, F_{});
    layer(std::integral_constant<int, 4>{}, PB, PA, F_{});
    layer(std::integral_constant<int, 3>{}, PA, PB, F_{});
    layer(std::integral_constant<int, 2>{}, PB, PA, F_{});
    layer(std::integral_constant<int, 1>{}, PA, PB, F_{});
    // ---- the last pass of dy_0 (out-blocks 6, 7): nothing left to ride under
    {
        const RowIO io0 = make_rowio(a.dsave + SL.h, 256, MP, blk, lane);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = 16 * b + r;
                const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)hbits[0][3], 31 - p, 1);
                const float x = __uint_as_float(__float_as_uint(acc1[b][r]) & keep);
                DMN_ACT_STORE_B32(f2u(x), io0.rs, run_off(0, r), (int)(io0.soff + (6 + b) * 4096), DMN_STORE_AUX);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the last (landing-zone) fetches
}

}  // namespace

extern "C" int dmnerf_grad_scale(const float* d_graw, int64_t n, float* d_scale4, void* stream) {
    if (!d_graw || !d_scale4 || n < 1) return dmn_fail(DMNERF_E_ARG, "grad_scale: bad argument");
    const unsigned blocks = (unsigned)((n + 4095) / 4096 < 512 ? (n + 4095) / 4096 : 512);
    hipLaunchKernelGGL(grad_scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_graw, n, d_scale4);
    return dmn_check_launch("grad_scale");
}

extern "C" int dmnerf_mlp_bwd_data_f16(const float* d_blob_t_f16, int ins_num, const float* d_save, const float* d_graw, int64_t M,
                                       float* d_dsave, float* d_graw_t, const float* d_scale, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data_f16: ins_num %d unsupported", ins_num);
    if (M < 0 || M > DMNERF_MAX_TRAIN_SAMPLES) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data_f16: M=%lld outside [0,%lld]", (long long)M, (long long)DMNERF_MAX_TRAIN_SAMPLES);
    if (M == 0) return DMNERF_OK;
    if (!d_blob_t_f16 || !d_save || !d_graw || !d_dsave) return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data_f16: null pointer");
    BwdHArgs a{};
    a.blob = d_blob_t_f16; a.L = make_layout(ins_num); a.LT = make_layout_t(ins_num); a.S = make_f16_layout_t(ins_num);
    a.save = d_save; a.graw = d_graw; a.dsave = d_dsave; a.graw_t = d_graw_t; a.scale = d_scale; a.M = M;
    const int64_t nblk = (M + 31) / 32;
    dim3 g((unsigned)((nblk + 3) / 4)), b(256);
    constexpr size_t lds_bytes = (size_t)(F16_RING_FLOATS + TAB_T_FLOATS) * sizeof(float);
#define DMN_LAUNCH(OBI_)                                                                                          \
    {                                                                                                            \
        static DmnOncePerDevice once;                                                                                 \
        if (hipError_t e_ = once.run([] { return hipFuncSetAttribute((const void*)mlp_bwd_f16_kernel<OBI_>,              \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }); e_ != hipSuccess) \
            return dmn_fail_hip(e_, "mlp_bwd_data_f16: hipFuncSetAttribute");                                       \
        hipLaunchKernelGGL(mlp_bwd_f16_kernel<OBI_>, g, b, lds_bytes, (hipStream_t)stream, a);                        \
    }
    switch (a.L.OBI) {
        case 1: DMN_LAUNCH(1) break;
        case 2: DMN_LAUNCH(2) break;
        case 3: DMN_LAUNCH(3) break;
        case 4: DMN_LAUNCH(4) break;
        default: return dmn_fail(DMNERF_E_ARG, "mlp_bwd_data_f16: unsupported logit count C=%d", a.L.C);
    }
#undef DMN_LAUNCH
    return dmn_check_launch("mlp_bwd_data_f16");
}
