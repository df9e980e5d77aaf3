// Microbenchmark (GPU box): would a 16-sample wave tile hold the f32 MFMA rate?
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_16x16.hip -o build_exp/mfma_16x16
//
// The fused MLP kernels carry 32 samples per wave on v_mfma_f32_32x32x2_f32 (64 cycles, 64 weights per instruction): a workgroup
// of 4 waves = 128 samples streams one 64 KiB weight quarter per 256 MFMAs = 16 384 cycles: 64 ds_read_b128 per wave (one per
// 4 MFMAs = 256 cycles) and 16 LDS-DMA pieces of 1 KiB per wave, one barrier.  A 16-sample wave on v_mfma_f32_16x16x4_f32
// (32 cycles, also 64 weights per instruction) halves a workgroup's trip through the network (64 samples, 8192 cycles per
// quarter), which would let the 384-ray shard finish in 5 half-rounds instead of 3 rounds -- IF the same 64 reads, 16 DMA pieces
// and the barrier fit into half the cycles: an LDS read per 128 cycles per wave, the weight stream at 2x the rate.
//
// Body = 256 MFMAs per wave with, pinned by sched_barriers: R: one ds_read_b128 in the first 8 gaps of every 32 MFMAs (= 64),
// D: n buffer_load..lds pieces of 1 KiB (the product's DMA form), B: a barrier, V: n VALU ops per MFMA gap (the epilogue's share).
// Kill criterion (written before the measurement): the 16x16x4 stream with reads + 16 DMA + barrier must sustain >= 0.90 of the
// plain f32 MFMA rate, else the kernel is not worth building.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <utility>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LAS __attribute__((address_space(3)))

template <class F, int... I>
__device__ __forceinline__ void sf_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sf_impl(f, std::make_integer_sequence<int, N>{}); }

// SMALL = 1: v_mfma_f32_16x16x4_f32 (16 accumulators of 4 registers); 0: v_mfma_f32_32x32x2_f32 (16 accumulators of 16)
template <int SMALL, int READS, int DMAS, int BARRIER, int VALU, int STORE = 0>
__global__ __launch_bounds__(256) void mix(float* out, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    using acc_t = std::conditional_t<SMALL, f32x4, f32x16>;
    acc_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (acc_t)(0.f);
    f32x4 a[2][8];
    for (int i = 0; i < 8; ++i) { a[0][i] = (f32x4)(1e-3f * tid); a[1][i] = (f32x4)(2e-3f * tid); }
    float b = blockIdx.x * 1e-3f + 1.f, vv = tid;
    const unsigned s0 = (unsigned)(unsigned long long)(LAS const void*)lds + lane * 16;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const unsigned voff = lane * 16;
    const unsigned long long gb = (unsigned long long)src + wu * 1024;
    const unsigned ub_lo = __builtin_amdgcn_readfirstlane((unsigned)gb), ub_hi = __builtin_amdgcn_readfirstlane((unsigned)(gb >> 32));
    const unsigned long long ub = ((unsigned long long)ub_hi << 32) | ub_lo;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ub, 0, 1 << 28, 0x00020000);
    __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc((void*)ub, 4, 64, (1 << 23));   // stride 4 + ADD_TID_ENABLE
    // STORE 1: the 16-sample wave's activation rows in the [32-sample block][row][32] layout the backward kernels read: lane
    // (q = lane / 16, n = lane % 16) writes feature row 4 q + i at sample n -> four 64-byte runs per instruction, address in a VGPR
    const int srow = ((lane >> 4) * 4) * 128 + (lane & 15) * 4 + (wu & 1) * 64;
    for (int it = 0; it < iters; ++it) {
        if (DMAS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
        sfor<8>([&](auto gc) {
            constexpr int gl = decltype(gc)::value;
            if (READS) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[gl & 1][i]));
            }
            __builtin_amdgcn_sched_barrier(0);
            sfor<32>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int M = gl * 32 + i;
                if constexpr (READS && i < 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(gl + 1) & 1][i]) : "v"(s0), "n"(((gl * 8 + i) & 63) * 1024));
                if constexpr (DMAS > 0 && M % (256 / (DMAS > 0 ? DMAS : 1)) == 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LAS void*)(lds + 16384 + wu * 256 + (M / (256 / (DMAS > 0 ? DMAS : 1))) * 1024), 16, voff,
                                                             (M / (256 / (DMAS > 0 ? DMAS : 1))) * 4096, 0, 0);
                for (int v = 0; v < VALU; ++v) vv = fmaxf(vv * 1.0001f, 0.5f);
                if constexpr (STORE == 1 && M % 16 == 0)        // 16 VGPR-addressed dword stores per body (64 per 256 x 256 layer)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a[gl & 1][i & 7][0]), rs, srow + (M / 16) * 2048, (it & 63) * 65536 + 4194304, 0);
                if constexpr (STORE == 2 && M % 8 == 0)         // 32 TID-addressed dword stores per body (128 per layer: the 32-sample kernel)
                    asm volatile("buffer_store_dword %0, off, %1, %2 offset:%3" :: "v"(a[gl & 1][i & 7][0]), "s"(rst), "s"((it & 63) * 65536 + 4194304 + (M / 8) * 256), "n"(0) : "memory");
                if constexpr (SMALL)
                    acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gl & 1][i & 7][i >> 3 & 3], b, acc[i & 15], 0, 0, 0);
                else
                    acc[i & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gl & 1][i & 7][i >> 3 & 3], b, acc[i & 15], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    float s = vv;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < (SMALL ? 4 : 16); ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double base_tf[2] = {0, 0};

template <int SMALL, int READS, int DMAS, int BARRIER, int VALU, int STORE = 0>
void run(const char* name) {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    int grid = p.multiProcessorCount, iters = SMALL ? 4000 : 2000;
    float *out, *src;
    (void)hipMalloc(&out, grid * 256 * 4); (void)hipMalloc(&src, 1 << 28); (void)hipMemset(src, 0, 1 << 20);
    auto k = mix<SMALL, READS, DMAS, BARRIER, VALU, STORE>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<grid, 256, 147456>>>(out, src, 400);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k<<<grid, 256, 147456>>>(out, src, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop_per_mfma = SMALL ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2;
    const double tf = (double)grid * 4 * iters * 256.0 * flop_per_mfma / best * 1e-9;
    const double ideal = SMALL ? 8192.0 : 16384.0;
    const double cyc = best * 1e-3 * 2.4e9 / iters;
    if (!READS && !DMAS && !BARRIER && !VALU && !STORE) base_tf[SMALL] = tf;
    printf("%-58s %8.3f ms %7.1f TFLOP/s = %.3f of 157.3 | %.3f of the plain stream | %6.0f cycles per 256 MFMAs at 2.4 GHz (ideal %5.0f, +%5.0f)\n",
           name, best, tf, tf / 157.3, base_tf[SMALL] > 0 ? tf / base_tf[SMALL] : 1.0, cyc, ideal, cyc - ideal);
    fflush(stdout);
    (void)hipFree(out); (void)hipFree(src);
}

int main() {
    run<0, 0, 0, 0, 0>("32x32x2  mfma only (warm-up)");
    run<0, 0, 0, 0, 0>("32x32x2  mfma only");
    run<0, 1, 0, 0, 0>("32x32x2  + 64 ds_read_b128");
    run<0, 1, 16, 1, 0>("32x32x2  + 64 reads + 16 DMA + vmcnt + barrier (the kernel)");
    run<0, 1, 16, 1, 1>("32x32x2  ... + 1 VALU per gap");
    run<1, 0, 0, 0, 0>("16x16x4  mfma only");
    run<1, 1, 0, 0, 0>("16x16x4  + 64 ds_read_b128 (one per 128 cycles)");
    run<1, 0, 16, 0, 0>("16x16x4  + 16 DMA (weight stream at 2x the rate)");
    run<1, 1, 16, 1, 0>("16x16x4  + 64 reads + 16 DMA + vmcnt + barrier");
    run<1, 1, 16, 1, 1>("16x16x4  ... + 1 VALU per gap (same epilogue work per MFMA)");
    run<1, 1, 8, 1, 0>("16x16x4  + 64 reads + 8 DMA + vmcnt + barrier");
    run<0, 1, 16, 1, 0, 2>("32x32x2  kernel + 32 TID-addressed row stores (training forward)");
    run<1, 1, 16, 1, 0, 1>("16x16x4  kernel + 16 VGPR-addressed row stores (training forward)");
    return 0;
}
