// split_f16.h -- the "f16x2" primitives shared by the opt-in split-f16 kernels (mlp_f16_impl.h, mlp_bwd_f16.hip, wgrad_f16.hip):
//   x ~ hi + lo,  hi = f16(x) (round toward zero: v_cvt_pkrtz_f16_f32 packs a pair in one instruction and saturates at 65504
//   instead of overflowing to inf),  lo = f16(x - hi)  (x - hi is exact in f32; lo may be an f16 SUBNORMAL, which both the
//   conversion and v_mfma_f32_32x32x16_f16 keep: scripts/micro/mfma_f16.hip, profiles/r03/mfma_f16_micro_r03a.txt);
//   |x - hi - lo| <= max(2^-21 |x|, 2^-24).
// A product of two f32 becomes THREE f16 MFMA products (hi hi + hi lo + lo hi, each exact in f32) accumulated in f32: the
// dropped lo lo term is below 2^-20 |w x|.  Emulated through the whole network against float64 (scripts/split_emulate.py):
// max |d raw| / (1 + |raw|) = 4.2e-7 (f32 GEMMs: 2.2e-7, bf16x3: 1.7e-7) -- f32 class at half the MFMAs of bf16x3.
#pragma once
#include <hip/hip_runtime.h>

#include "mlp_common.h"

using namespace dmn;

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f16x8 as_bh(const unsigned* w) {
    const u32x4 v = {w[0], w[1], w[2], w[3]};
    return __builtin_bit_cast(f16x8, v);
}
__device__ __forceinline__ f16x8 as_ah(const f32x4& v) { return __builtin_bit_cast(f16x8, v); }

// (x0, x1) -> the two f16-pair words.  Four VALU instructions per pair: the residual x - hi is ONE v_fma_mix_f32 each (it
// reads the f16 half of the packed word directly: -hi * 1.0 + x, exact).
__device__ __forceinline__ void split_pair_f16(float x0, float x1, unsigned& whi, unsigned& wlo) {
    float r0, r1;
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %3, %4\n\t"
                 "v_fma_mix_f32 %1, -%0, 1.0, %3 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %2, -%0, 1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(whi), "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1));
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(wlo) : "v"(r0), "v"(r1));
}

}  // namespace
