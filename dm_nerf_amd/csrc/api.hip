// api.hip -- error plumbing and the whole-path entry point of the C ABI.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/dmnerf_hip.h"
#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

int dmn_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int dmn_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return DMNERF_OK;
    return dmn_fail(DMNERF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

int dmn_fail_hip(hipError_t e, const char* what) {
    (void)hipGetLastError();                                 // clear the sticky error: it is being reported here
    return dmn_fail(DMNERF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e != hipSuccess ? e : hipErrorUnknown));
}

extern "C" int dmnerf_abi_version(void) { return DMNERF_ABI_VERSION; }
extern "C" const char* dmnerf_last_error(void) { return g_err; }

extern "C" int dmnerf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// dm_nerf inference (networks/render.py:31-96): every stage enqueued on one stream, no sync.
extern "C" int dmnerf_render_rays_fwd(const dmnerf_render_args* a, void* stream) {
    if (!a) return dmn_fail(DMNERF_E_ARG, "render_rays_fwd: null args");
    if (a->N == 0 && a->S >= 3 && a->n_imp >= 1) return DMNERF_OK;      // an empty chunk: its buffers may be null
    if (!a->d_blob_coarse || !a->d_blob_fine || !a->d_rays_o || !a->d_rays_d || !a->d_z_in || !a->d_u ||
        !a->d_z_coarse || !a->d_raw_coarse || !a->d_rgb_coarse || !a->d_depth_coarse || !a->d_ins_coarse ||
        !a->d_z_fine || !a->d_raw_fine || !a->d_rgb_fine || !a->d_depth_fine || !a->d_ins_fine || !a->d_weights_ws)
        return dmn_fail(DMNERF_E_ARG, "render_rays_fwd: null pointer in args");
    const int64_t N = a->N;
    const int S = a->S, SF = a->S + a->n_imp, C = a->ins_num + 1;
    if (N < 0 || S < 3 || a->n_imp < 1) return dmn_fail(DMNERF_E_ARG, "render_rays_fwd: bad N=%lld S=%d n_imp=%d", (long long)N, S, a->n_imp);
    if (N == 0) return DMNERF_OK;
    int rc;
    // stratified jitter (render.py:40-47) or pass-through copy of the coarse grid
    if (a->d_t_rand) {
        if ((rc = dmnerf_stratify(a->d_z_in, a->d_t_rand, N, S, a->d_z_coarse, stream))) return rc;
    } else if (a->d_z_coarse != a->d_z_in) {
        if (hipMemcpyAsync(a->d_z_coarse, a->d_z_in, sizeof(float) * N * S, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return dmn_check_launch("render_rays_fwd: z copy");
    }
    // coarse network + compositing (render.py:49-63)
    auto mlp = a->fused_heads == 3 ? dmnerf_mlp_fwd_rays_f16 : a->fused_heads == 2 ? dmnerf_mlp_fwd_rays_split : (a->fused_heads ? dmnerf_mlp_fwd_rays_fused : dmnerf_mlp_fwd_rays);
    if ((rc = mlp(a->d_blob_coarse, a->ins_num, a->d_rays_o, a->d_rays_d, a->d_z_coarse, N, S, a->d_raw_coarse, stream))) return rc;
    if ((rc = dmnerf_composite_fwd(a->d_raw_coarse, a->d_z_coarse, a->d_rays_d, N, S, C, a->d_rgb_coarse, a->d_weights_ws,
                                   a->d_depth_coarse, a->d_ins_coarse, stream))) return rc;
    // hierarchical resampling + merge (render.py:66-70)
    if ((rc = dmnerf_importance_resample(a->d_z_coarse, a->d_weights_ws, a->d_u, a->u_row_stride, N, S, a->n_imp, a->d_z_fine, nullptr, stream))) return rc;
    // fine network + compositing (render.py:71-86)
    if (a->ev_fine_mlp_begin) (void)hipEventRecord((hipEvent_t)a->ev_fine_mlp_begin, (hipStream_t)stream);
    if ((rc = mlp(a->d_blob_fine, a->ins_num, a->d_rays_o, a->d_rays_d, a->d_z_fine, N, SF, a->d_raw_fine, stream))) return rc;
    if (a->ev_fine_mlp_end) (void)hipEventRecord((hipEvent_t)a->ev_fine_mlp_end, (hipStream_t)stream);
    if ((rc = dmnerf_composite_fwd(a->d_raw_fine, a->d_z_fine, a->d_rays_d, N, SF, C, a->d_rgb_fine, a->d_weights_ws,
                                   a->d_depth_fine, a->d_ins_fine, stream))) return rc;
    return DMNERF_OK;
}
