// mlp_fwd_embedded.hip -- DM_NeRF.forward on pre-embedded rows [M, 90] (kernel: mlp_fwd_impl.h)
// (one translation unit per entry point: the 4 logit-block instantiations of a variant compile in parallel with the others)
#include "mlp_fwd_impl.h"

extern "C" int dmnerf_mlp_fwd_embedded(const float* d_blob, int ins_num, const float* d_x, int64_t M,
                                       float* d_raw, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_embedded: ins_num %d unsupported", ins_num);
    if (M < 0) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_embedded: M < 0");
    if (M == 0) return DMNERF_OK;      // an empty batch is legal (and has null data pointers)
    if (!d_blob || !d_x || !d_raw) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_embedded: null pointer");
    MlpArgs a{};
    a.blob = d_blob; a.L = make_layout(ins_num); a.x = d_x; a.raw = d_raw; a.M = M; a.S = 1;
    return launch<true, false>(a, (hipStream_t)stream);
}
