// Microbenchmark (GPU box): sustained v_mfma_f32_32x32x2_f32 rate with nothing else in the loop.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o build_exp/mfma_peak
// Prints TFLOP/s for 1, 2 waves per SIMD and the shader clock implied by s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void spin(float* out, long long* clk, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x16)(0.f);
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int NACC>
void run(int wgs_per_cu, const char* name) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount, grid = cus * wgs_per_cu, iters = 20000;
    float* out; long long* clk;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&clk, grid * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin<NACC><<<grid, 256>>>(out, clk, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    spin<NACC><<<grid, 256>>>(out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flop = (double)grid * 4 * iters * 4.0 * NACC * 4096.0;
    printf("%s: %d WGs/CU x 4 waves, %d accumulators: %.3f ms  %.1f TFLOP/s   s_memtime/s_memrealtime = %.3f (x100 MHz = %.0f MHz)\n",
           name, wgs_per_cu, NACC, ms, flop / ms * 1e-9, (double)h[0] / h[1], 100.0 * h[0] / h[1]);
    hipFree(out); hipFree(clk);
}

int main() {
    run<4>(1, "4acc"); run<8>(1, "8acc"); run<16>(1, "16acc"); run<4>(2, "4acc-2wg"); run<8>(2, "8acc-2wg");
    return 0;
}
