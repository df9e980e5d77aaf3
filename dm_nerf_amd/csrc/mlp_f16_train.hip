// mlp_f16_train.hip -- OPT-IN training forward on the split-f16 MFMA path (args.mfma_split = "f16x2" with grad enabled): the
// inference kernel of mlp_f16_impl.h with SAVE = true -- it also writes the f32 activations (pe, de, h_0..h_7, g1, g2) and the
// 1-bit ReLU masks into the SaveLayout workspace that the backward kernels consume (the f32 ones of mlp_bwd.hip / wgrad.hip as
// well as their split twins).  The stores and the mask packing ride in the MFMA gaps with the rest of the epilogue (four gaps
// per element pair); this translation unit is built with -DDMN_STORE_AUX=2 (csrc/Makefile): the 7.8 GB of row stores per fine launch
// are `nt`, which took the kernel from 2.75 to 2.36 ms (docs/EXPERIMENTS.md section 8).  The forward is the fused-heads function, i.e. exactly the form the re-associated backward
// differentiates; values are f32-class (2e-7 against float64) but not the bitwise fmaf chain of the default path.
#include "mlp_f16_impl.h"

#ifdef DMN_F16_TRACE
static long long* g_f16_train_trace = nullptr;
extern "C" void dmnerf_f16_train_set_trace(long long* d_trace) { g_f16_train_trace = d_trace; }
#endif

extern "C" int dmnerf_mlp_fwd_rays_train_f16(const float* d_blob_f16, int ins_num, const float* d_rays_o, const float* d_rays_d,
                                             const float* d_z, int64_t N, int S, float* d_raw, float* d_save, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train_f16: ins_num %d unsupported", ins_num);
    if (N < 0 || S < 1) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train_f16: bad N=%lld S=%d", (long long)N, S);
    if (N == 0) return DMNERF_OK;
    if (!d_blob_f16 || !d_rays_o || !d_rays_d || !d_z || !d_raw || !d_save) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train_f16: null pointer");
    F16Args a{};
    a.blob = d_blob_f16; a.S = make_f16_layout(ins_num);
    a.rays_o = d_rays_o; a.rays_d = d_rays_d; a.z = d_z; a.raw = d_raw; a.save = d_save; a.M = N * S; a.Sr = S;
#ifdef DMN_F16_TRACE
    a.trace = g_f16_train_trace;
#endif
    if (a.M > DMNERF_MAX_TRAIN_SAMPLES)
        return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train_f16: %lld samples per launch exceed %lld; split the batch", (long long)a.M, (long long)DMNERF_MAX_TRAIN_SAMPLES);
    const int64_t nblk = (a.M + 31) / 32;
    const int64_t grid = (nblk + 3) / 4;
    constexpr size_t lds_bytes = (size_t)F16_LDS_FLOATS * sizeof(float);
#define DMN_LAUNCH(OBX_)                                                                                                   \
    {                                                                                                                     \
        static DmnOncePerDevice once;                                                                                 \
        if (hipError_t e_ = once.run([] { return hipFuncSetAttribute((const void*)mlp_f16_kernel<OBX_, true>,              \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }); e_ != hipSuccess) \
            return dmn_fail_hip(e_, "mlp_fwd_rays_train_f16: hipFuncSetAttribute");                                       \
        hipLaunchKernelGGL((mlp_f16_kernel<OBX_, true>), dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, a);    \
    }
    switch (a.S.OBX) {
        case 1: DMN_LAUNCH(1) break;
        case 2: DMN_LAUNCH(2) break;
        case 4: DMN_LAUNCH(4) break;
        default: return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train_f16: unsupported logit count C=%d", a.S.C);
    }
#undef DMN_LAUNCH
    return dmn_check_launch("mlp_fwd_rays_train_f16");
}
