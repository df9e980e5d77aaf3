// mlp_fwd.hip -- inference entry point of the fused PE + DM-NeRF MLP kernel on rays (kernel: mlp_fwd_impl.h)
// (one translation unit per entry point: the 4 logit-block instantiations of a variant compile in parallel with the others)
#include "mlp_fwd_impl.h"

#ifdef DMN_FWD_TRACE
long long* g_dmn_fwd_trace = nullptr;
extern "C" int dmnerf_debug_fwd_trace(long long* p) { g_dmn_fwd_trace = p; return 0; }
#endif

extern "C" int dmnerf_mlp_fwd_rays(const float* d_blob, int ins_num, const float* d_rays_o,
                                   const float* d_rays_d, const float* d_z, int64_t N, int S,
                                   float* d_raw, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays: ins_num %d unsupported", ins_num);
    if (N < 0 || S < 1) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays: bad N=%lld S=%d", (long long)N, S);
    if (N == 0) return DMNERF_OK;
    if (!d_blob || !d_rays_o || !d_rays_d || !d_z || !d_raw) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays: null pointer");
    MlpArgs a{};
    a.blob = d_blob; a.L = make_layout(ins_num); a.rays_o = d_rays_o; a.rays_d = d_rays_d; a.z = d_z;
    a.raw = d_raw; a.M = N * S; a.S = S;
#ifdef DMN_FWD_TRACE
    a.trace = g_dmn_fwd_trace;
#endif
    return launch<false, false>(a, (hipStream_t)stream);
}
