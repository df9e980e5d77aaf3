// mlp_common.h -- device helpers shared by the fused MLP forward and backward kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

namespace dmn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Weight stream: buffer loads through one wave-uniform descriptor (SGPRs) with the per-lane
// part (lane*16 B) in a single voffset VGPR and the segment position in the scalar offset, so
// no 64-bit per-load address ever occupies VGPRs (flat addressing spilled ~300 address pairs).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t make_rsrc(const float* p, int64_t n_floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), /*stride*/ 0, (int)(n_floats * 4), 0x00020000);
}

__device__ __forceinline__ f32x4 ldw(rsrc_t r, int voff, int soff_bytes) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff_bytes, 0);
    return __builtin_bit_cast(f32x4, v);
}

// NOTE: __builtin_bit_cast applied directly to an ext_vector ELEMENT expression reads element 0
// (hipcc 7.2): always go through a scalar copy.
__device__ __forceinline__ unsigned f2u(float x) { return __float_as_uint(x); }

struct RowIO {
    rsrc_t rs;
    int voff;        // ((4*half)*M + m) * 4 bytes
    unsigned rowb;   // M * 4: bytes per row
    bool valid;
};

// acc[ob] += Wseg[ob-block rows, k] * B[k, samples] for NKG*4 k-pairs.  B lives in registers in
// accumulator layout: k-pair p is B[p >> 4][p & 15].  `seg` = float offset of the segment.
template <int NKG, int OB, int NB, bool STORE_B = false>
__device__ __forceinline__ void gemm_seg(rsrc_t rs, int seg, const f32x16 (&B)[NB],
                                         f32x16 (&acc)[OB], int voff, const RowIO* sio = nullptr) {
    static_assert(NB * 16 >= NKG * 4, "B operand too small");
    // The segment is streamed front to back; its position lives in ONE scalar register that is
    // bumped every 4 KiB (imm offsets cover 0..3 KiB).  The empty asm makes the running value
    // opaque, otherwise the scheduler materialises every `base + const` offset up front and
    // spills hundreds of SGPRs (seen as v_writelane/v_readlane storms and scratch traffic).
    int so = seg * 4;
    asm volatile("" : "+s"(so));
    // STORE_B (training): the B operand (this GEMM's input activations, in accumulator layout) is
    // written feature-major to HBM *while it is being consumed*: 4 dword stores per k-group, i.e.
    // one store per 8 MFMAs, instead of a 128-store burst at the layer boundary whose
    // acknowledgements the next weight loads would have to wait behind (vmcnt is in-order).
    // Stores are unconditional (no exec-mask branches inside the MFMA stream): rows are padded to a
    // multiple of 32 samples so tail lanes write padding, and a descriptor with num_records = 0
    // turns a whole call into no-ops through the hardware bounds check.
    int svo = 0, step1 = 0, step5 = 0;
    if constexpr (STORE_B) {
        svo = sio->voff; step1 = (int)sio->rowb; step5 = (int)(5 * sio->rowb);
        asm volatile("" : "+v"(svo));
    }
#pragma unroll
    for (int g = 0; g < NKG; ++g) {
        f32x4 a[OB];
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
            const int lin = g * OB + ob;                 // KiB index inside the segment
            a[ob] = ldw(rs, voff, so + (lin & 3) * 1024);
            if ((lin & 3) == 3) { so += 4096; asm volatile("" : "+s"(so)); }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int p = g * 4 + kk;
            if constexpr (STORE_B) {
                __builtin_amdgcn_raw_buffer_store_b32(f2u(B[p >> 4][p & 15]), sio->rs, svo, 0, 0);
                svo += ((p & 3) == 3) ? step5 : step1;
            }
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
                acc[ob] = mfma32(a[ob][kk], B[p >> 4][p & 15], acc[ob]);
            }
        }
    }
}

// acc[ob][r] = bias of row 32ob + (r&3) + 8(r>>2) + 4half; hoff = half * 64 bytes.
template <int OB>
__device__ __forceinline__ void init_bias(rsrc_t rs, int seg, f32x16 (&acc)[OB], int hoff) {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = ldw(rs, hoff, (seg + ob * 32 + q * 4) * 4);
            acc[ob][4 * q + 0] = v[0]; acc[ob][4 * q + 1] = v[1];
            acc[ob][4 * q + 2] = v[2]; acc[ob][4 * q + 3] = v[3];
        }
    }
}

__device__ __forceinline__ f32x16 relu16(f32x16 v) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = fmaxf(v[i], 0.f);
    return r;
}

// Encoding of one 3-vector in the k-pair order of layout.h::pefeat: lanes 0-31 take the sin
// slot (and x, z), lanes 32-63 the cos slot (and y, pad).  sin/cos are ocml's full-range f32
// routines (arguments reach 2^9 * |x|: no fast-math approximations here).
template <int L, int NV>
__device__ __forceinline__ void encode(const float (&v)[3], f32x16 (&out)[NV], int half) {
    static_assert(NV * 16 >= 2 + 3 * L, "encoding registers too small");
#pragma unroll
    for (int i = 0; i < NV; ++i) out[i] = (f32x16)(0.f);
    out[0][0] = half ? v[1] : v[0];
    out[0][1] = half ? 0.f : v[2];
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p = 2 + 3 * k + c;
            const float arg = v[c] * (float)(1 << k);   // exact: power of two (dm_nerf.py:25,31)
            float s, co;
            sincosf(arg, &s, &co);
            out[p >> 4][p & 15] = half ? co : s;
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


// ---- feature-major activation tensors [rows][M] (training): row = feature, column = sample ----
// A wave's accumulator block b, register r holds feature 32b + crow(r, half) of sample m: for a
// fixed (b, r) lanes 0-31 are 32 consecutive samples of one row and lanes 32-63 of the row 4 below,
// i.e. two fully coalesced 128-byte segments per instruction, with no transpose.
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

__device__ __forceinline__ RowIO make_rowio(const float* base, int rows, int64_t M, int64_t m, int half, bool valid) {
    RowIO io;
    io.rs = uniform_rsrc(base, (int64_t)rows * M);
    io.voff = (int)((4 * (int64_t)half * M + m) * 4);
    io.rowb = (unsigned)(M * 4);
    io.valid = valid;
    return io;
}

// Rows are visited in increasing order with a running per-lane byte offset (one VALU add per
// access): row steps are +1,+1,+1,+5 (crow pattern), so no per-row scalar offset is ever live
// (hoisted row*M products spilled SGPRs -> scratch in the first version).
template <int NB>
__device__ __forceinline__ void store_rows(const RowIO& io, const f32x16 (&v)[NB]) {
    int vo = io.voff;
    asm volatile("" : "+v"(vo));     // opaque per call: identical offset chains of different calls must not be CSE'd into ~128 live VGPRs
    const int step1 = (int)io.rowb, step5 = (int)(5 * io.rowb);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_raw_buffer_store_b32(f2u(v[b][r]), io.rs, vo, 0, 0);
            vo += ((r & 3) == 3) ? step5 : step1;
        }
    }
}

template <int NB>
__device__ __forceinline__ void load_rows(const RowIO& io, f32x16 (&v)[NB]) {
    int vo = io.voff;
    asm volatile("" : "+v"(vo));
    const int step1 = (int)io.rowb, step5 = (int)(5 * io.rowb);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[b][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(io.rs, vo, 0, 0));
            vo += ((r & 3) == 3) ? step5 : step1;
        }
    }
}

// Encoded inputs are saved in Embedder.embed's own column order (rows of a [3+6L][M] matrix).
template <int L, int NV>
__device__ __forceinline__ void store_encoded_rows(const float* base, int64_t M, int64_t m, int half, bool valid,
                                                   const f32x16 (&e)[NV]) {
    (void)valid;                                               // rows are padded: tail lanes write padding
    rsrc_t rs = uniform_rsrc(base, (int64_t)(3 + 6 * L) * M);
    const unsigned rowb = (unsigned)(M * 4);
    const int v1 = (int)(((int64_t)half * M + m) * 4);        // rows 0/1 (x, y)
    const int v3 = (int)((3 * (int64_t)half * M + m) * 4);    // sin row + 3 = cos row
    __builtin_amdgcn_raw_buffer_store_b32(f2u(e[0][0]), rs, v1, 0, 0);
    if (half == 0) __builtin_amdgcn_raw_buffer_store_b32(f2u(e[0][1]), rs, v1 + (int)(2 * rowb), 0, 0);
    int vo = v3 + (int)(3 * rowb);                            // row 3 + 6k + c (+3 for the cos half)
    asm volatile("" : "+v"(vo));
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p = 2 + 3 * k + c;
            __builtin_amdgcn_raw_buffer_store_b32(f2u(e[p >> 4][p & 15]), rs, vo, 0, 0);
            vo += (c == 2) ? (int)(4 * rowb) : (int)rowb;
        }
    }
}

}  // namespace dmn
