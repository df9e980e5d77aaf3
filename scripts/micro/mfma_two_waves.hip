// Microbenchmark (GPU box): does a second wave per SIMD hide one wave's VALU work under the other's MFMAs?
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_two_waves.hip -o build_exp/mfma_two_waves
// Every wave runs `iters` bodies of 256 v_mfma_f32_32x32x2_f32 (4 accumulators, round robin) followed by a BURST of
// NV independent v_max_i32 (the shape of a layer epilogue).  One workgroup per CU (100 KiB of LDS requested), 256
// threads = one wave per SIMD or 512 threads = two.  Reported: cycles per MFMA per SIMD against the 64 of the pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int THREADS, int NV>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x16)(0.f);
    int v[64];
    for (int i = 0; i < 64; ++i) v[i] = tid * 3 + i - 90;
    float a = 1e-3f * tid, b = 1.f + blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 256; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[j & 63]) : "v"(it));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    int t = 0;
    for (int i = 0; i < 64; ++i) t += v[i];
    if (s == 12345.678f || t == 42) out[tid] = s + t + lds[tid];
}

template <int THREADS, int NV>
void run(const char* name) {
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 4000, cus = 256;
    hipFuncSetAttribute((const void*)k<THREADS, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<THREADS, NV>), dim3(cus), dim3(THREADS), 100 * 1024, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = THREADS / 256.0;
    const double mfma_per_simd = 256.0 * iters * waves_per_simd;
    const double tf = 2.0 * 2048 * 256.0 * iters * (THREADS / 64) * cus / (ms * 1e-3) / 1e12;
    printf("%-44s %7.3f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)  ns per MFMA per SIMD %.2f\n", name, ms, tf, 100 * tf / 157.3,
           ms * 1e6 / mfma_per_simd);
    hipFree(out);
}

int main() {
    run<256, 0>("1 wave/SIMD, MFMA only (warm-up)");
    run<256, 0>("1 wave/SIMD, MFMA only");
    run<256, 64>("1 wave/SIMD, + 64 VALU burst per 256 MFMA");
    run<256, 256>("1 wave/SIMD, + 256 VALU burst per 256 MFMA");
    run<512, 0>("2 waves/SIMD, MFMA only");
    run<512, 64>("2 waves/SIMD, + 64 VALU burst per 256 MFMA");
    run<512, 256>("2 waves/SIMD, + 256 VALU burst per 256 MFMA");
    return 0;
}
