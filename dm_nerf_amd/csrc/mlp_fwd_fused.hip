// mlp_fwd_fused.hip -- inference on rays with the fused-heads blob (opt-in; kernel: mlp_fwd_impl.h)
// (one translation unit per entry point: the 4 logit-block instantiations of a variant compile in parallel with the others)
#include "mlp_fwd_impl.h"

extern "C" int dmnerf_mlp_fwd_rays_fused(const float* d_blob_fused, int ins_num, const float* d_rays_o,
                                         const float* d_rays_d, const float* d_z, int64_t N, int S,
                                         float* d_raw, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_fused: ins_num %d unsupported", ins_num);
    if (N < 0 || S < 1) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_fused: bad N=%lld S=%d", (long long)N, S);
    if (N == 0) return DMNERF_OK;
    if (!d_blob_fused || !d_rays_o || !d_rays_d || !d_z || !d_raw) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_fused: null pointer");
    MlpArgs a{};
    a.blob = d_blob_fused; a.L = make_layout(ins_num, true); a.rays_o = d_rays_o; a.rays_d = d_rays_d; a.z = d_z;
    a.raw = d_raw; a.M = N * S; a.S = S;
#ifdef DMN_FWD_TRACE
    a.trace = nullptr;
#endif
    return launch<false, false, true>(a, (hipStream_t)stream);
}
