// Microbenchmark (GPU box): feasibility of a split-bf16 ("bf16x3") MLP kernel.  v_mfma_f32_32x32x16_bf16 stream, one
// wave per SIMD, with 0 or 1 ds_read_b128 (a fresh A operand) per MFMA -- the split path needs one 16-byte A
// operand per MFMA, 4x the LDS traffic per MFMA-cycle of the f32 kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/mfma_bf16.hip -o build_exp/mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <utility>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define LAS __attribute__((address_space(3)))

template <class F, int... I>
__device__ __forceinline__ void sf_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sf_impl(f, std::make_integer_sequence<int, N>{}); }

template <int READS, int WGS>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) lds[i] = 1e-3f * i;
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x16)(0.f);
    f32x4 a[2][8];
    for (int i = 0; i < 8; ++i) { a[0][i] = (f32x4)(1e-3f * tid); a[1][i] = (f32x4)(2e-3f * tid); }
    f32x4 bv = (f32x4)(1.f + blockIdx.x * 1e-3f);
    const unsigned s0 = (unsigned)(unsigned long long)(LAS const void*)lds + lane * 16;
    for (int it = 0; it < iters; ++it) {
        sfor<16>([&](auto gc) {                       // 16 groups x 8 MFMAs = 128 MFMAs per body
            constexpr int gl = decltype(gc)::value;
            if (READS) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[gl & 1][i]));
            }
            __builtin_amdgcn_sched_barrier(0);
            sfor<8>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (READS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(gl + 1) & 1][i]) : "v"(s0), "n"(((gl * 8 + i) & 63) * 1024));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[gl & 1][i]), __builtin_bit_cast(bf16x8, bv), acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int READS, int WGS>
void run(const char* name) {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    int grid = p.multiProcessorCount * WGS, iters = 4000;
    float* out; (void)hipMalloc(&out, grid * 256 * 4);
    auto k = spin<READS, WGS>;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<grid, 256, 65536>>>(out, 400);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<<<grid, 256, 65536>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * 128.0 * 32768.0;
    printf("%-44s %.3f ms  %7.1f TFLOP/s (dense bf16 peak ~2500)\n", name, ms, flop / ms * 1e-9);
    fflush(stdout);
    (void)hipFree(out);
}

int main() {
    run<0, 1>("bf16 32x32x16 only, 1 wave/SIMD (warm-up)");
    run<0, 1>("bf16 32x32x16 only, 1 wave/SIMD");
    run<1, 1>("+ 1 ds_read_b128 per MFMA, 1 wave/SIMD");
    run<0, 2>("bf16 32x32x16 only, 2 waves/SIMD");
    run<1, 2>("+ 1 ds_read_b128 per MFMA, 2 waves/SIMD");
    return 0;
}
