// wgrad_split.hip -- OPT-IN weight-gradient kernel on the split-bf16 MFMA path (training with args.mfma_split = True / "bf16x3"):
// wgrad_split_impl.h's schedule with three planes of packed bf16 pairs (truncation split: exact, split_bf16.h) and six
// v_mfma_f32_32x32x16_bf16 per 16-sample step of a tile pair (hi hi, hi mid, hi lo, mid hi, mid mid, lo hi): 96 NBA NBB MFMA
// cycles per 32-sample chunk instead of 256 NBA NBB.  Measured (docs/EXPERIMENTS.md section 8): a 32-sample chunk of a 256 x 256 job
// 7 025 -> 4 436 ns (its MFMAs alone: 2 560 ns at 2.4 GHz); the skinny jobs were HBM- / hand-over-bound already and do not move.
#include "split_bf16.h"
#include "wgrad_split_impl.h"

namespace {

struct ModeBf16x3 {
    static constexpr int NP = 3, NT = 6, NTMP = 2;
    static constexpr int B_WAIT = 6;                   // slots between the last B read of a step and the first unit that uses one
    static constexpr int term_a(int t) { constexpr int A[NT] = {0, 0, 0, 1, 1, 2}; return A[t]; }      // planes of product t
    static constexpr int term_b(int t) { constexpr int B[NT] = {0, 1, 2, 0, 1, 0}; return B[t]; }
    // unit U = 2 q + stage of a block's pair q: stage 0 = hi word + first residuals (5 VALU), stage 1 = mid and lo words (6).  The
    // pins keep a unit in ITS slot (without them the compiler sinks the work to its first use).
    template <int U>
    __device__ static __forceinline__ void unit(const f32x4 (&r)[2], unsigned (&P)[NP][4], float (&t)[4][NTMP]) {
        constexpr int q = U >> 1;
        if constexpr ((U & 1) == 0) {
            const float x0 = r[q >> 1][2 * (q & 1)], x1 = r[q >> 1][2 * (q & 1) + 1];
            const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
            P[0][q] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
            t[q][0] = x0 - __uint_as_float(u0 & 0xffff0000u);
            t[q][1] = x1 - __uint_as_float(u1 & 0xffff0000u);
            asm volatile("" : "+v"(P[0][q]), "+v"(t[q][0]), "+v"(t[q][1]));
        } else {
            const unsigned v0 = __float_as_uint(t[q][0]), v1 = __float_as_uint(t[q][1]);
            P[1][q] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
            const float q0 = t[q][0] - __uint_as_float(v0 & 0xffff0000u), q1 = t[q][1] - __uint_as_float(v1 & 0xffff0000u);
            P[2][q] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
            asm volatile("" : "+v"(P[1][q]), "+v"(P[2][q]));
        }
    }
    __device__ static __forceinline__ f32x16 mfma(const unsigned (&pa)[4], const unsigned (&pb)[4], const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b(pa), as_b(pb), c, 0, 0, 0);
    }
};

__global__ __launch_bounds__(256) void wgrad_split_kernel(const WgArgs a) { wgrad_split_body<ModeBf16x3>(a); }

}  // namespace

extern "C" int dmnerf_mlp_bwd_weights_split(const float* d_save, const float* d_dsave, const float* d_graw_t, int64_t M,
                                            const void* d_jobs, int n_jobs, const void* d_outs, int n_outs,
                                            const float* d_params_flat, int ins_num, float* d_part, float* d_grad_flat, void* stream) {
    static DmnOncePerDevice once;
    return wgrad_split_launch(wgrad_split_kernel, once, "mlp_bwd_weights_split", d_save, d_dsave, d_graw_t, M, d_jobs, n_jobs, d_outs, n_outs, d_params_flat, ins_num,
                                          d_part, d_grad_flat, nullptr, stream);
}
