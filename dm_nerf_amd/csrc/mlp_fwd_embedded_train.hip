// mlp_fwd_embedded_train.hip -- DM_NeRF.forward on pre-embedded rows [M, 90] WITH saved activations: the training-mode
// forward of callers that embed the points themselves (networks/dm_nerf.py:80-106 is differentiable; mesh / third-party
// code calls the model directly).  Same kernel template (mlp_fwd_impl.h); the workspace it fills is the one
// dmnerf_mlp_bwd_data / dmnerf_mlp_bwd_weights consume.
#include "mlp_fwd_impl.h"

extern "C" int dmnerf_mlp_fwd_embedded_train(const float* d_blob, int ins_num, const float* d_x, int64_t M,
                                             float* d_raw, float* d_save, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_embedded_train: ins_num %d unsupported", ins_num);
    if (M < 0) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_embedded_train: M < 0");
    if (M == 0) return DMNERF_OK;
    if (!d_blob || !d_x || !d_raw || !d_save) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_embedded_train: null pointer");
    MlpArgs a{};
    a.blob = d_blob; a.L = make_layout(ins_num); a.x = d_x; a.raw = d_raw; a.save = d_save; a.M = M; a.S = 1;
#ifdef DMN_FWD_TRACE
    a.trace = nullptr;
#endif
    return launch<true, true>(a, (hipStream_t)stream);
}
