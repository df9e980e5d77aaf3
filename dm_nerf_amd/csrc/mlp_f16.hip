// mlp_f16.hip -- opt-in split-f16 ("f16x2") inference entry point + the two-plane weight packer (kernel: mlp_f16_impl.h)
#include "mlp_f16_impl.h"

namespace {

// ---- packer: stream words (two f16 each) from the flat f32 parameters; planes by round-to-nearest (offline: free)
__device__ __forceinline__ unsigned plane_bits_f16(float x, int plane) {
    const _Float16 hi = (_Float16)x;
    if (plane == 0) return (unsigned)__builtin_bit_cast(unsigned short, hi);
    const _Float16 lo = (_Float16)(x - (float)hi);
    return (unsigned)__builtin_bit_cast(unsigned short, lo);
}

__global__ void pack_f16_kernel(const float* __restrict__ flat, const int* __restrict__ idx, unsigned* __restrict__ words, int64_t n_words) {
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * blockDim.x) {
        unsigned out = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int id = idx[2 * w + e];
            if (id >= 0) out |= plane_bits_f16(flat[id & 0x0fffffff], id >> 28) << (16 * e);
        }
        words[w] = out;
    }
}

}  // namespace

extern "C" int dmnerf_pack_f16(const float* d_flat, const int32_t* d_idx, float* d_stream_words, int64_t n_words, void* stream) {
    if (!d_flat || !d_idx || !d_stream_words || n_words <= 0) return dmn_fail(DMNERF_E_ARG, "pack_f16: bad argument");
    const unsigned blocks = (unsigned)((n_words + 255) / 256 < 8192 ? (n_words + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_flat, (const int*)d_idx, (unsigned*)d_stream_words, n_words);
    return dmn_check_launch("pack_f16");
}

#ifdef DMN_F16_TRACE
static long long* g_f16_trace = nullptr;
extern "C" void dmnerf_f16_set_trace(long long* d_trace) { g_f16_trace = d_trace; }
#endif

extern "C" int dmnerf_mlp_fwd_rays_f16(const float* d_blob_f16, int ins_num, const float* d_rays_o, const float* d_rays_d,
                                       const float* d_z, int64_t N, int S, float* d_raw, void* stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_f16: ins_num %d unsupported", ins_num);
    if (N < 0 || S < 1) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_f16: bad N=%lld S=%d", (long long)N, S);
    if (N == 0) return DMNERF_OK;
    if (!d_blob_f16 || !d_rays_o || !d_rays_d || !d_z || !d_raw) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_f16: null pointer");
    F16Args a{};
    a.blob = d_blob_f16; a.S = make_f16_layout(ins_num);
    a.rays_o = d_rays_o; a.rays_d = d_rays_d; a.z = d_z; a.raw = d_raw; a.M = N * S; a.Sr = S;
#ifdef DMN_F16_TRACE
    a.trace = g_f16_trace;
#endif
    const int64_t nblk = (a.M + 31) / 32;
    const int64_t grid = (nblk + 3) / 4;
    if (grid > 0x7fffffffLL) return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_f16: too many samples");
    constexpr size_t lds_bytes = (size_t)F16_LDS_FLOATS * sizeof(float);
#define DMN_LAUNCH(OBX_)                                                                                                   \
    {                                                                                                                     \
        static DmnOncePerDevice once;                                                                                 \
        if (hipError_t e_ = once.run([] { return hipFuncSetAttribute((const void*)mlp_f16_kernel<OBX_>,              \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); }); e_ != hipSuccess) \
            return dmn_fail_hip(e_, "mlp_fwd_rays_f16: hipFuncSetAttribute");                                       \
        hipLaunchKernelGGL(mlp_f16_kernel<OBX_>, dim3((unsigned)grid), dim3(256), lds_bytes, (hipStream_t)stream, a);    \
    }
    switch (a.S.OBX) {
        case 1: DMN_LAUNCH(1) break;
        case 2: DMN_LAUNCH(2) break;
        case 4: DMN_LAUNCH(4) break;
        default: return dmn_fail(DMNERF_E_ARG, "mlp_fwd_rays_f16: unsupported logit count C=%d", a.S.C);
    }
#undef DMN_LAUNCH
    return dmn_check_launch("mlp_fwd_rays_f16");
}
