// split_bf16.h -- the "bf16x3" primitives shared by the opt-in split-bf16 kernels (mlp_split_impl.h, mlp_bwd_split.hip,
// wgrad_split.hip):  x = hi + mid + lo by truncation (exact for any f32), planes kept as packed bf16 pairs; a product of two f32
// becomes six bf16 MFMA products (hi hi, hi mid, mid hi, hi lo, mid mid, lo hi) accumulated in f32.
#pragma once
#include <hip/hip_runtime.h>

#include "mlp_common.h"

using namespace dmn;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 as_b(const unsigned* w) {
    const u32x4s v = {w[0], w[1], w[2], w[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 as_a(const f32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// (x0, x1) -> the three bf16-pair words of the truncation split (x = hi + mid + lo exactly); the two subtractions of a
// stage are one v_pk_add_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& whi, unsigned& wmid, unsigned& wlo) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    whi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const f32x2 x = {x0, x1};
    const f32x2 h = {__uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 & 0xffff0000u)};
    const f32x2 r = x - h;
    const unsigned v0 = __float_as_uint(r[0]), v1 = __float_as_uint(r[1]);
    wmid = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const f32x2 m = {__uint_as_float(v0 & 0xffff0000u), __uint_as_float(v1 & 0xffff0000u)};
    const f32x2 q = r - m;
    wlo = __builtin_amdgcn_perm(__float_as_uint(q[1]), __float_as_uint(q[0]), 0x07060302u);
}

// The same split with single-lane f32 subtractions: beside MFMAs of a one-wave-per-SIMD kernel a v_pk_add_f32 costs ~13
// cycles beyond its issue slot (MI355X_MICROARCH.md, "price of one filler beside MFMAs"), two v_sub_f32 cost their two slots.
// (The translation unit must be built with -fno-slp-vectorize or the compiler re-packs them.)
__device__ __forceinline__ void split_pair_scalar(float x0, float x1, unsigned& whi, unsigned& wmid, unsigned& wlo) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    whi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    wmid = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - __uint_as_float(v0 & 0xffff0000u), q1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    wlo = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

}  // namespace
