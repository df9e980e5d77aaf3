// mlp_common.h -- device helpers shared by the fused MLP forward and backward kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "layout.h"

namespace dmn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// relu as ONE integer max on the bit pattern (negative floats are negative ints; -0 -> +0): fmaxf costs a
// second v_max_f32 (sNaN quieting), and every VALU instruction is paid for in MFMA time.
__device__ __forceinline__ float relu1(float x) {
    const int xi = (int)__float_as_uint(x);
    return __uint_as_float((unsigned)(xi > 0 ? xi : 0));
}
__device__ __forceinline__ f32x16 relu16(f32x16 v) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = relu1(v[i]);
    return r;
}

// sin / cos of v * 2^k for k = 0..L-1 with ONE shared range reduction in double precision:
//   t = v / (2 pi) (double: |error| <= |t| 2^-53), and for each k the angle in revolutions is t * 2^k, an
//   exact scaling; g = frac-to-nearest(t 2^k + q/4) in [-1/2, 1/2] is exact as well (q = 0: sin, q = 1: cos,
//   cos x = sin(x + pi/2)); fold to [-1/4, 1/4] and evaluate sin(2 pi g) by its degree-15 Taylor polynomial in
//   double (truncation 6e-12 at pi/2), round once to float.  Arguments reach 2^9 |x| ~ 7700 rad: the result is
//   within 1/2 ulp + 2e-12 of the true value of the SAME float argument the reference feeds to sin / cos
//   (x * freq is exact for power-of-two freq, dm_nerf.py:25,31), cheaper and tighter than a full-range sincosf.
__device__ __forceinline__ double rev_of(float v) { return (double)v * 0.15915494309189535; }      // 1 / (2 pi)

__device__ __forceinline__ double sin_rev_d(double t, int k, int quarter) {
    double g = t * (double)(1 << k) + 0.25 * (double)quarter;
    g = g - __builtin_rint(g);
    const double ag = __builtin_fabs(g);
    const double gf = ag > 0.25 ? 0.5 - ag : ag;
    const double x = gf * 6.283185307179586;
    const double x2 = x * x;
    double p = 1.0 / 1307674368000.0;                    // 1/15!
    p = __builtin_fma(p, x2, -1.0 / 6227020800.0);       // 1/13!
    p = __builtin_fma(p, x2, 1.0 / 39916800.0);
    p = __builtin_fma(p, x2, -1.0 / 362880.0);
    p = __builtin_fma(p, x2, 1.0 / 5040.0);
    p = __builtin_fma(p, x2, -1.0 / 120.0);
    p = __builtin_fma(p, x2, 1.0 / 6.0);
    const double s = x - x * x2 * p;                     // x (1 - x2/6 + x4/120 - ...)
    return __builtin_copysign(s, g);
}
__device__ __forceinline__ float sin_rev(double t, int k, int quarter) { return (float)sin_rev_d(t, k, quarter); }

// Encoding of one 3-vector in the k-pair order of layout.h::pefeat: lanes 0-31 take the sin
// slot (and x, z), lanes 32-63 the cos slot (and y, pad).
// Per coordinate ONE sin / cos pair is evaluated (frequency 1) and the higher octaves follow from the double-
// angle recurrence in double precision, s' = 2 s c, c' = 1 - 2 s^2: the absolute error grows like 2^k x 1e-16
// (<= 1e-13 at k = 9), far below half an f32 ulp of values in [-1, 1], at 5 instead of ~19 f64 instructions per
// output (the encoding is pure VALU work in front of the first MFMA).
template <int L, int NV>
__device__ __forceinline__ void encode(const float (&v)[3], f32x16 (&out)[NV], int half) {
    static_assert(NV * 16 >= 2 + 3 * L, "encoding registers too small");
#pragma unroll
    for (int i = 0; i < NV; ++i) out[i] = (f32x16)(0.f);
    out[0][0] = half ? v[1] : v[0];
    out[0][1] = half ? 0.f : v[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double t = rev_of(v[c]);
        double sn = sin_rev_d(t, 0, 0), cs = sin_rev_d(t, 0, 1);
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const int p = 2 + 3 * k + c;
            out[p >> 4][p & 15] = (float)(half ? cs : sn);
            const double s_old = sn, t2 = s_old + s_old;
            sn = t2 * cs;                                    // sin 2a = 2 sin a cos a
            cs = __builtin_fma(-t2, s_old, 1.0);             // cos 2a = 1 - 2 sin^2 a
        }
    }
}

// Same registers filled from a pre-embedded row (DM_NeRF.forward called directly on [M,90]).
template <int L, int NV>
__device__ __forceinline__ void load_encoded(const float* __restrict__ e, f32x16 (&out)[NV], int half) {
#pragma unroll
    for (int i = 0; i < NV; ++i) out[i] = (f32x16)(0.f);
    out[0][0] = e[half];
    out[0][1] = half ? 0.f : e[2];
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p = 2 + 3 * k + c;
            out[p >> 4][p & 15] = e[3 + 6 * k + 3 * half + c];
        }
    }
}


// ---- block-major activation tensors (training) ------------------------------------------------
// Every saved tensor with R rows (features) is stored as [block of 32 samples][R rows][32 samples]:
//   addr(blk, row, j) = ((blk * R + row) * 32 + j) floats.
// A wave's accumulator block b, register r holds feature 32b + crow(r, half) of sample j: for a
// fixed (b, r) lanes 0-31 write the 128-byte row segment of feature f and lanes 32-63 that of
// f + 4, and ALL the stores of a layer land in one contiguous R*128-byte region (32 KiB for 256
// rows): one TLB entry, a handful of DRAM pages.  (The first version used feature-major [R][M]
// rows 3 MB apart: every store / every wgrad row hit a different page and the kernels ran at a
// quarter of their speed.)  The weight-gradient kernel reads the same regions as contiguous tiles.
#ifndef DMN_STORE_AUX
#define DMN_STORE_AUX 0   /* cache policy bits of the activation stores (2 = nt) */
#endif
// Diagnostic variant (make variant NAME=nostore FLAGS=-DDMN_NO_ACT_STORES; never shipped): the activation / gradient
// row stores compiled out, to separate what the store INSTRUCTIONS cost from what their HBM traffic does to the clock.
#ifdef DMN_NO_ACT_STORES
#define DMN_ACT_STORE_B32(...) ((void)0)
#else
#define DMN_ACT_STORE_B32(...) __builtin_amdgcn_raw_buffer_store_b32(__VA_ARGS__)
#endif
// Accumulator-layout tensors (h, f, q, g1, g2 and their gradients) are stored with TID-ADDRESSED stores:
// the descriptor has stride 4 + ADD_TID_ENABLE, so lane l of a `buffer_store_dword v, off, rsrc, soffset
// offset:imm` writes base + soffset + imm + 4 l -- one contiguous 256-byte run per instruction and NO
// address VGPR.  A store with an address VGPR costs ~27 cycles of a one-wave MFMA stream, this form none
// (scripts/micro/mfma_mix.hip), and a 256 -> 256 layer saves 128 of them.  For that, the two features a
// register holds (f in lanes 0-31, f + 4 in lanes 32-63) must be adjacent 128-byte rows, so inside every
// group of 8 features the MEMORY row order is 0,4,1,5,2,6,3,7:
//     mem_row(f) = (f & ~7) + 2 (f & 3) + ((f >> 2) & 1),    feature(mem_row) = layout.h::row_feature()
// (the weight-gradient kernel maps its output rows/columns back; nothing else reads these tensors).
struct RowIO {
    rsrc_t rs;        // stride 4, ADD_TID_ENABLE, 64 records (the record count does not bound the lanes: a wave
                      // beyond the end of the batch is an exact duplicate of the last block's wave instead)
    unsigned soff;    // byte offset of this wave's 32-sample block: blk * R * 128 (uniform)
};

// NOTE: __builtin_bit_cast applied directly to an ext_vector ELEMENT expression reads element 0
// (hipcc 7.2): always go through a scalar copy.
__device__ __forceinline__ unsigned f2u(float x) { return __float_as_uint(x); }

// Descriptor from provably wave-uniform inputs: without the readfirstlane the compiler keeps the
// (uniform) pointer in VGPRs under SGPR pressure and wraps every access in a waterfall loop.
__device__ __forceinline__ rsrc_t uniform_rsrc(const float* base, int64_t n_floats) {
    const unsigned long long p = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane((unsigned)(n_floats * 4));
    float* q = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)nb, 0x00020000);
}

// base: tensor start; R: its row count; blk: this wave's block (Mp is unused: kept for the call sites' symmetry); ob0: first
// out-block (32 rows) of the tensor this RowIO writes -- dg1 and dg2 share one 256-row tensor (ob0 = 0 / 4) so that the weight-
// gradient kernel reads them as ONE A operand against h_7.
__device__ __forceinline__ RowIO make_rowio(const float* base, int R, int64_t Mp, int64_t blk, int lane, int ob0 = 0) {
    RowIO io;
    const unsigned long long p = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    float* q = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
    // with ADD_TID_ENABLE the DATA_FORMAT bits of word 3 are stride[17:14]: they stay 0
    io.rs = __builtin_amdgcn_make_buffer_rsrc(q, 4, 64, 1 << 23);
    (void)Mp;
#ifdef DMN_ACT_STORE_HOT      /* diagnostic: every store lands in 8 blocks that stay in L2 -- the instructions without their HBM traffic */
    blk &= 7;
#endif
    io.soff = __builtin_amdgcn_readfirstlane((unsigned)(blk * R * 128 + ob0 * 4096));
    (void)lane;
    return io;
}

// register r of out-block b -> byte offset of its 256-byte run inside the block (lanes 0-31: feature
// 32 b + (r & 3) + 8 (r >> 2), lanes 32-63: that + 4)
__device__ __forceinline__ constexpr int run_off(int b, int r) { return (32 * b + 8 * (r >> 2) + 2 * (r & 3)) * 128; }

template <int NB>
__device__ __forceinline__ void store_rows(const RowIO& io, const f32x16 (&v)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            DMN_ACT_STORE_B32(f2u(v[b][r]), io.rs, run_off(0, r), (int)(io.soff + b * 4096), DMN_STORE_AUX);
    }
}

// Encoded inputs are saved in Embedder.embed's own column order (rows of a [blk][3+6L][32] tensor).
template <int L, int NV>
__device__ __forceinline__ void store_encoded_rows(const float* base, int64_t Mp, int64_t blk, int lane,
                                                   const f32x16 (&e)[NV]) {
    constexpr int R = 3 + 6 * L;
    const int half = lane >> 5, j = lane & 31;
    rsrc_t rs = uniform_rsrc(base, (int64_t)R * Mp);
    const int v0 = (int)((blk * R * 32 + j) * 4);
    __builtin_amdgcn_raw_buffer_store_b32(f2u(e[0][0]), rs, v0 + half * 128, 0, 0);           // rows 0 / 1: x, y
    if (half == 0) __builtin_amdgcn_raw_buffer_store_b32(f2u(e[0][1]), rs, v0 + 2 * 128, 0, 0);  // row 2: z
    const int v3 = v0 + 3 * half * 128;                                                        // sin row, +3 rows = cos row
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p = 2 + 3 * k + c;
            __builtin_amdgcn_raw_buffer_store_b32(f2u(e[p >> 4][p & 15]), rs, v3 + ((3 + 6 * k + c) * 128) % 4096, ((3 + 6 * k + c) * 128) / 4096 * 4096, 0);
        }
    }
}


// ==========================================================================================
// Weight streaming through LDS (the "loader-less" engine of the fused MLP kernels)
// ==========================================================================================
// LDS = [ring: 2 slots x 64 KiB][table: 16 KiB].  The weight stream (layout.h) is consumed one
// 64 KiB quarter at a time; quarter q lives in slot q & 1.  All four waves of the workgroup issue
// the LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction, 16 per wave and quarter) for
// quarter q+1 during the first fourth of quarter q, and read their A operands with ds_read_b128, so
//   * no VMEM load result is ever waited for inside the MFMA stream, and every DMA is OLDER than the
//     activation stores issued after it;
//   * the L2 -> CU weight traffic drops 4x (one copy per workgroup instead of one per wave).
// The hand-over of a quarter (vmcnt(0) + s_barrier) happens at the start of the previous quarter's last
// k-group: see gemm_quarter / ws_handover below.
constexpr int SLOT_FLOATS = QUARTER_FLOATS;
constexpr int RING_FLOATS = 2 * SLOT_FLOATS;
constexpr int LDS_FLOATS = RING_FLOATS + TAB_FLOATS;       // 147 456 bytes
constexpr int DMA_PER_QUARTER = 16;                        // LDS-DMA instructions per wave per quarter (1 KiB each)

#define DMN_GAS __attribute__((address_space(1)))
#define DMN_LAS __attribute__((address_space(3)))

struct WStream {
    rsrc_t rs;            // the whole blob as a raw buffer (uniform)
    unsigned voff;        // lane*16 + wave*1024: this lane's bytes of piece 0 of a quarter
    float* ring;          // LDS ring base
    int wave;             // wave id in the workgroup (uniform)
    unsigned off;         // byte offset from the blob start of the next quarter to fetch (uniform)
    int cslot;            // ring slot of the next quarter to consume
    f32x4 pre[8];         // A operands of the next quarter's first k-group (read during this quarter's last group)
};

__device__ __forceinline__ void ws_init(WStream& ws, const float* blob, int64_t blob_floats, float* ring, int lane, int wave,
                                        int64_t stream_float_off) {
    ws.rs = uniform_rsrc(blob, blob_floats);
    ws.wave = __builtin_amdgcn_readfirstlane(wave);
    ws.voff = (unsigned)(lane * 16 + ws.wave * 1024);
    ws.ring = ring;
    ws.off = __builtin_amdgcn_readfirstlane((unsigned)(stream_float_off * 4));
    ws.cslot = 0;
}

// One 1-KiB piece (i = 0..15) of the next quarter: wave w fetches pieces 4 i + w.  MUBUF form (descriptor +
// one constant 32-bit VGPR offset + SGPR offset): NO per-piece vector ALU -- any VALU instruction of the
// wave takes its cycles from the MFMA stream (measured: 16 pieces addressed through 64-bit VGPR pointers
// cost ~900 cycles per quarter, this form ~0; scripts/micro/mfma_mix.hip).
__device__ __forceinline__ void ws_fetch_piece(const WStream& ws, int i) {
    float* dst = ws.ring + (ws.cslot ^ 1) * SLOT_FLOATS + ws.wave * 256 + i * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ws.rs, (DMN_LAS void*)dst, 16, (int)ws.voff, (int)(ws.off + i * 4096), 0, 0);
}
__device__ __forceinline__ void ws_fetch(WStream& ws) {
#pragma unroll
    for (int i = 0; i < DMA_PER_QUARTER; ++i) ws_fetch_piece(ws, i);
    ws.off += QUARTER_FLOATS * 4;
}

// Prologue: quarter 0 into slot 0 (nothing is being consumed yet).
__device__ __forceinline__ void ws_fetch_first(WStream& ws) {
    ws.cslot = 1;
    ws_fetch(ws);
    ws.cslot = 0;
}

// ---- hand-scheduled LDS operand reads --------------------------------------------------------------
// With one wave per SIMD nothing hides a stall of that wave, and the compiler's own placement of the
// A-operand reads (right before their s_waitcnt) and of the LDS-DMA burst (16 in 5 MFMA gaps) cost ~10 %
// of the MFMA rate.  The reads are therefore issued through inline asm (invisible to the compiler's
// waitcnt insertion), waited for by hand, and every MFMA is pinned in place by a sched_barrier.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(DMN_LAS const void*)p;
}
// Register class of the asm-read operand tiles, chosen per translation unit in the Makefile: "v" or "a" (an MFMA
// takes its A operand from either file).  The compiler believes an asm read has completed when it has been issued; if
// the register allocator then decides that the value should live in the OTHER file it copies a register whose read
// is still in flight.  Asking for the class the allocator wants anyway removes the copy; scripts/check_asm_hazard.py
// proves on the shipped ISA that none is left (the build fails otherwise).  CARRY selects the class of the hand-over
// tiles (WStream::pre), whose live range spans the code between two quarters.
#ifndef DMN_TILE_RC
#define DMN_TILE_RC "v"
#endif
#ifndef DMN_CARRY_RC
#define DMN_CARRY_RC DMN_TILE_RC
#endif
template <int OFF, bool CARRY = false>
__device__ __forceinline__ void lds_read16_async(f32x4& v, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_read_b128 offset field");
    if constexpr (CARRY) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=" DMN_CARRY_RC(v) : "v"(addr), "n"(OFF));
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=" DMN_TILE_RC(v) : "v"(addr), "n"(OFF));
}
// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>) -- the
// index is a constant expression inside f (literal instruction offsets; no address arithmetic for the
// compiler to hoist out of the layer loop and spill)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// wait until at most N LDS operations are outstanding, then hand the registers back to the compiler
template <int N, bool CARRY = false, int NV>
__device__ __forceinline__ void lds_wait(f32x4 (&v)[NV]) {
    __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));           // lgkmcnt(N) only (gfx9 encoding)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if constexpr (CARRY) asm volatile("" : "+" DMN_CARRY_RC(v[i]));
        else asm volatile("" : "+" DMN_TILE_RC(v[i]));
    }
}

// Quarter hand-over.  Quarter q + 1 is complete in the other ring slot once every wave has seen its own
// DMA pieces land (vmcnt(0)) and passed the barrier; the barrier also releases quarter q's slot for the DMA
// of q + 2, because a wave only arrives after its last LDS read of q has returned.  It sits at the START of
// quarter q's last k-group (whose operands are already in registers), so the first operands of q + 1 are
// read under that group's MFMAs and no LDS latency is exposed at the quarter boundary.
template <int NYOUNGER = 0>
__device__ __forceinline__ void ws_handover() {
    // vmcnt retires in order: "at most NYOUNGER outstanding" = everything older than the NYOUNGER activation
    // stores issued after this quarter's DMA pieces has landed
    __builtin_amdgcn_s_waitcnt(0x0F70 | (NYOUNGER & 15) | ((NYOUNGER >> 4) << 14));
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct NoSide {
    __device__ __forceinline__ void operator()(int) const {}
};

// Very first quarter of the kernel: wait for it (and for the LDS table), then read its first operands.
template <int OB>
__device__ __forceinline__ void ws_prime(WStream& ws, int lane) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned s0 = lds_addr(ws.ring + ws.cslot * SLOT_FLOATS) + lane * 16;
    static_for<OB>([&](auto ic) {
        constexpr int ob = decltype(ic)::value;
        lds_read16_async<ob * 1024, true>(ws.pre[ob], s0);
    });
    lds_wait<0, true>(ws.pre);    // (once per workgroup; see the end of gemm_quarter)
}

// One quarter's worth of a GEMM segment: k-groups [G0, G0 + NG) of a segment with OB out-blocks, A
// operands from the LDS slot (group-local index), B from registers (accumulator layout).  NEXT_OB = out-
// blocks of the quarter that follows in the stream (0: this is the kernel's last quarter).
// Schedule per k-group (4 OB MFMAs, k-major): the OB reads of the NEXT group ride in the first OB MFMA
// gaps (register double buffer), the 16 DMA pieces of the next quarter are spread over the first fourth of
// the quarter, one per gap; the last group starts with the hand-over and reads the next quarter's first
// operands into ws.pre.  NSIDE "side" operations side(0) .. side(NSIDE-1) (training: TID-addressed activation
// stores, which cost nothing when they are SPREAD over the MFMA stream but stall it when issued as a burst)
// ride in every third gap between the DMA pieces and the last group.
template <int G0, int NG, int OB, int NEXT_OB, bool ZERO = false, int NSIDE = 0, int NB, class Side = NoSide>
__device__ __forceinline__ void gemm_quarter(WStream& ws, const f32x16 (&B)[NB], f32x16 (&acc)[OB], int lane, Side&& side = Side()) {
    static_assert(NB * 16 >= (G0 + NG) * 4, "B operand too small");
    static_assert(NG * OB <= 64, "more than one quarter");
    constexpr int GM = 4 * OB;                       // MFMAs per k-group
    constexpr int Q = NG * GM;                       // MFMAs in this call
    constexpr int P = (Q / 4) / DMA_PER_QUARTER > 0 ? (Q / 4) / DMA_PER_QUARTER : 1;   // gaps between DMA pieces (first quarter of the call)
    static_assert(OB + (DMA_PER_QUARTER - 1) * P < Q - GM, "DMA pieces must be issued before the last group");
    static_assert(NG >= 2 && NEXT_OB <= GM && NEXT_OB <= 8, "hand-over does not fit the last group");
    constexpr int S0 = OB + DMA_PER_QUARTER * P;      // first gap of the side operations
    constexpr int SP = NSIDE > 0 ? ((Q - GM - S0) / NSIDE >= 3 ? 3 : ((Q - GM - S0) / NSIDE >= 1 ? (Q - GM - S0) / NSIDE : 1)) : 1;
    static_assert(NSIDE == 0 || (S0 + (NSIDE - 1) * SP < Q - GM && NSIDE <= 63), "side operations do not fit before the last group");
    const unsigned s0 = lds_addr(ws.ring + ws.cslot * SLOT_FLOATS) + lane * 16;
    const unsigned s1 = lds_addr(ws.ring + (ws.cslot ^ 1) * SLOT_FLOATS) + lane * 16;
    f32x4 a[2][OB];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) a[0][ob] = ws.pre[ob];
    static_for<NG>([&](auto gc) {
        constexpr int gl = decltype(gc)::value;
        lds_wait<0>(a[gl & 1]);
        if constexpr (gl == NG - 1 && NEXT_OB > 0) ws_handover<NSIDE>();
        __builtin_amdgcn_sched_barrier(0);
        static_for<GM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int kk = i / OB, ob = i % OB, M = gl * GM + i;
            constexpr int p = (G0 + gl) * 4 + kk;
            if constexpr (gl + 1 < NG && i < OB)
                lds_read16_async<((gl + 1) * OB + i) * 1024>(a[(gl + 1) & 1][i], s0);           // group gl + 1, out-block i
            if constexpr (gl == NG - 1 && i < NEXT_OB) lds_read16_async<i * 1024, true>(ws.pre[i], s1);  // next quarter, group 0
            if constexpr (M >= OB && (M - OB) % P == 0 && (M - OB) / P < DMA_PER_QUARTER) ws_fetch_piece(ws, (M - OB) / P);
            if constexpr (NSIDE > 0 && M >= S0 && (M - S0) % SP == 0 && (M - S0) / SP < NSIDE) side((M - S0) / SP);
            // ZERO: this quarter starts the GEMM -- C = 0 is an inline constant of the MFMA, no accumulator clear
            if constexpr (ZERO && G0 + gl == 0 && kk == 0) acc[ob] = mfma32(a[gl & 1][ob][kk], B[p >> 4][p & 15], (f32x16)(0.f));
            else acc[ob] = mfma32(a[gl & 1][ob][kk], B[p >> 4][p & 15], acc[ob]);
            __builtin_amdgcn_sched_barrier(0);
        });
    });
    // ws.pre was written by inline-asm reads the compiler believes to be synchronous: turn it into real values before
    // the caller's code between two quarters (epilogues, stores) gives the register allocator a reason to move it
    // (the reads were issued at the start of the last k-group, >= 4 OB MFMAs ago: this wait is free)
    if constexpr (NEXT_OB > 0) __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
    ws.off += QUARTER_FLOATS * 4;
    ws.cslot ^= 1;
}

// acc[ob][r] = bias of row 32ob + crow(r, half), from the LDS table
template <int OB>
__device__ __forceinline__ void init_bias_lds(const float* tab_seg, f32x16 (&acc)[OB], int half) {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
        const f32x4* b4 = reinterpret_cast<const f32x4*>(tab_seg + (ob * 2 + half) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = b4[q];
            acc[ob][4 * q + 0] = v[0]; acc[ob][4 * q + 1] = v[1];
            acc[ob][4 * q + 2] = v[2]; acc[ob][4 * q + 3] = v[3];
        }
    }
}

// ---- 1-bit ReLU masks (layout.h::BITS_WORDS_PER_BLOCK) ---------------------------------------------
// Element p = 32 w + i of the tensor is bit (31 - i) of word w: each element costs a compare and an
// add-with-carry (m = 2 m + [h > 0]); the NB/2 words are advanced round-robin so that consecutive VALU
// instructions are independent.
template <int NB>
__device__ __forceinline__ void pack_mask(const f32x16 (&h)[NB], unsigned (&m)[NB / 2]) {
#pragma unroll
    for (int w = 0; w < NB / 2; ++w) m[w] = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
#pragma unroll
        for (int w = 0; w < NB / 2; ++w) {
            const int p = 32 * w + i;
            const float x = h[p >> 4][p & 15];                          // relu backward: grad * (result > 0)
            asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m[w]) : "v"(x) : "vcc");
        }
    }
}

template <int NB>
__device__ __forceinline__ void apply_mask(f32x16 (&d)[NB], const unsigned (&m)[NB / 2], const f32x16 (&g)[NB]) {
#pragma unroll
    for (int p = 0; p < NB * 16; ++p) {
        // sign-extended 1-bit field = 0 or ~0: two VALU ops per register (bfe + and)
        const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)m[p >> 5], 31 - (p & 31), 1);
        d[p >> 4][p & 15] = __uint_as_float(__float_as_uint(g[p >> 4][p & 15]) & keep);
    }
}

// One register (k-pair numbering p = 16 b + r) of an accumulator-layout tensor.
template <int NB>
__device__ __forceinline__ void store_row_one(const RowIO& io, const f32x16 (&v)[NB], int p) {
    DMN_ACT_STORE_B32(f2u(v[p >> 4][p & 15]), io.rs, run_off(0, p & 15), (int)(io.soff + (p >> 4) * 4096), DMN_STORE_AUX);
}

// Stores registers [P0, P0 + NP) (k-pair numbering p = 16 b + r) of an accumulator-layout tensor.
template <int P0, int NP, int NB>
__device__ __forceinline__ void store_rows_part(const RowIO& io, const f32x16 (&v)[NB]) {
#pragma unroll
    for (int p = P0; p < P0 + NP; ++p)
        DMN_ACT_STORE_B32(f2u(v[p >> 4][p & 15]), io.rs, run_off(0, p & 15), (int)(io.soff + (p >> 4) * 4096), DMN_STORE_AUX);
}

}  // namespace dmn
