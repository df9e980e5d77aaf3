// wgrad_f16.hip -- OPT-IN weight-gradient kernel on the split-f16 MFMA path (training with args.mfma_split = "f16x2"):
// wgrad_split_impl.h's schedule with two planes of packed f16 pairs (split_f16.h: v_cvt_pkrtz_f16_f32 + v_fma_mix_f32, 4
// instructions per pair instead of 11) and THREE v_mfma_f32_32x32x16_f16 per 16-sample step of a tile pair (hi hi, hi lo, lo hi):
// 48 NBA NBB MFMA cycles per 32-sample chunk instead of 96 (bf16x3) / 256 (f32).  The dy operand carries the power-of-two factor
// of dmnerf_grad_scale (f16 range), which the second stage removes.  With the MFMA time halved again the 256 x 256 jobs sit on
// the HBM stream of their operands (64 KiB per chunk and workgroup: 2.7 us at 6.3 TB/s against 1.3 us of MFMA issue): the
// schedule only has to stay under that shadow.
#include "split_f16.h"
#include "wgrad_split_impl.h"

namespace {

struct ModeF16x2 {
    static constexpr int NP = 2, NT = 3, NTMP = 1;
    static constexpr int B_WAIT = 4;
    static constexpr int term_a(int t) { constexpr int A[NT] = {0, 0, 1}; return A[t]; }      // planes of product t
    static constexpr int term_b(int t) { constexpr int B[NT] = {0, 1, 0}; return B[t]; }
    // unit U = 2 q + stage of a block's pair q: stage 0 = hi word + the first residual, stage 1 = the second residual + lo word (2
    // instructions each; the residual x - hi is ONE v_fma_mix_f32 reading the f16 half of the packed word).  volatile asm keeps a
    // unit in ITS slot.
    template <int U>
    __device__ static __forceinline__ void unit(const f32x4 (&r)[2], unsigned (&P)[NP][4], float (&t)[4][NTMP]) {
        constexpr int q = U >> 1;
        const float x0 = r[q >> 1][2 * (q & 1)], x1 = r[q >> 1][2 * (q & 1) + 1];
        if constexpr ((U & 1) == 0) {
            asm volatile("v_cvt_pkrtz_f16_f32 %0, %2, %3\n\t"
                         "v_fma_mix_f32 %1, -%0, 1.0, %2 op_sel_hi:[1,0,0]" : "=&v"(P[0][q]), "=&v"(t[q][0]) : "v"(x0), "v"(x1));
        } else {
            float t1;
            asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(t1) : "v"(P[0][q]), "v"(x1));
            asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(P[1][q]) : "v"(t[q][0]), "v"(t1));
        }
    }
    __device__ static __forceinline__ f32x16 mfma(const unsigned (&pa)[4], const unsigned (&pb)[4], const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(as_bh(pa), as_bh(pb), c, 0, 0, 0);
    }
};

__global__ __launch_bounds__(256) void wgrad_f16_kernel(const WgArgs a) { wgrad_split_body<ModeF16x2>(a); }

}  // namespace

extern "C" int dmnerf_mlp_bwd_weights_f16(const float* d_save, const float* d_dsave, const float* d_graw_t, int64_t M,
                                          const void* d_jobs, int n_jobs, const void* d_outs, int n_outs,
                                          const float* d_params_flat, int ins_num, float* d_part, float* d_grad_flat, const float* d_scale, void* stream) {
    static DmnOncePerDevice once;
    // the dy operands carry the factor 2^s of dmnerf_grad_scale: the second stage multiplies by d_scale[1] = 2^-s
    return wgrad_split_launch(wgrad_f16_kernel, once, "mlp_bwd_weights_f16", d_save, d_dsave, d_graw_t, M, d_jobs, n_jobs, d_outs, n_outs, d_params_flat, ins_num,
                                         d_part, d_grad_flat, d_scale ? d_scale + 1 : nullptr, stream);
}
