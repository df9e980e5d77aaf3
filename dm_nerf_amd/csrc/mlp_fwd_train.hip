// mlp_fwd_train.hip -- training forward (saves activations + ReLU masks; kernel: mlp_fwd_impl.h)
// (one translation unit per entry point: the 4 logit-block instantiations of a variant compile in parallel with the others)
#include "mlp_fwd_impl.h"

extern "C" int64_t dmnerf_train_save_floats(int64_t M) { return M < 0 ? -1 : make_save_layout(M).total; }

extern "C" int dmnerf_mlp_fwd_rays_train(const float* d_blob, int ins_num, const float* d_rays_o,
                                         const float* d_rays_d, const float* d_z, int64_t N, int S,
                                         float* d_raw, float* d_save, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train: ins_num %d unsupported", ins_num);
    if (N < 0 || S < 1) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train: bad N=%lld S=%d", (long long)N, S);
    if (N == 0) return DMNERF_OK;
    if (!d_blob || !d_rays_o || !d_rays_d || !d_z || !d_raw || !d_save) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train: null pointer");
    MlpArgs a{};
    a.blob = d_blob; a.L = make_layout(ins_num); a.rays_o = d_rays_o; a.rays_d = d_rays_d; a.z = d_z;
    a.raw = d_raw; a.save = d_save; a.M = N * S; a.S = S;
#ifdef DMN_FWD_TRACE
    a.trace = g_dmn_fwd_trace;
#endif
    return launch<false, true>(a, (hipStream_t)stream);
}
