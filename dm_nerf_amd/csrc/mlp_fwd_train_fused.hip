// mlp_fwd_train_fused.hip -- OPT-IN training forward on the fused-heads blob (args.fuse_heads in training mode): the two
// activation-free feature linears folded into the hidden layers (562 432 instead of 693 504 MAC per sample), activations
// saved for the same backward (mlp_bwd.hip / wgrad.hip / heads.hip are in re-associated form already).  Results equal the
// layer-by-layer forward up to f32 re-association -- inside the 1e-5 (1 + |raw|) contract, not bit-equal to the default
// path, hence opt-in and never part of the headline numbers.  Kernel: mlp_fwd_impl.h.
#include "mlp_fwd_impl.h"

extern "C" int dmnerf_mlp_fwd_rays_train_fused(const float* d_blob_fused, int ins_num, const float* d_rays_o,
                                               const float* d_rays_d, const float* d_z, int64_t N, int S,
                                               float* d_raw, float* d_save, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train_fused: ins_num %d unsupported", ins_num);
    if (N < 0 || S < 1) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train_fused: bad N=%lld S=%d", (long long)N, S);
    if (N == 0) return DMNERF_OK;
    if (!d_blob_fused || !d_rays_o || !d_rays_d || !d_z || !d_raw || !d_save) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_train_fused: null pointer");
    MlpArgs a{};
    a.blob = d_blob_fused; a.L = make_layout(ins_num, true); a.rays_o = d_rays_o; a.rays_d = d_rays_d; a.z = d_z;
    a.raw = d_raw; a.save = d_save; a.M = N * S; a.S = S;
#ifdef DMN_FWD_TRACE
    a.trace = nullptr;
#endif
    return launch<false, true, true>(a, (hipStream_t)stream);
}
