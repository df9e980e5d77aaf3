// Microbenchmark (GPU box): what each non-MFMA instruction class costs a 1-wave-per-SIMD MFMA stream.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_mix.hip -o build_exp/mfma_mix
// Loop body = 256 v_mfma_f32_32x32x2_f32 (16 accumulators) per wave with, pinned by sched_barriers:
//   R: one ds_read_b128 in the first 8 gaps of every 32 MFMAs      D: n DMA pieces (global_load_lds 1 KiB), one per 8 gaps
//   B: one s_barrier per 256 MFMAs                                   V: n VALU ops per gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <utility>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))
#define LAS __attribute__((address_space(3)))

template <class F, int... I>
__device__ __forceinline__ void sf_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sf_impl(f, std::make_integer_sequence<int, N>{}); }

template <int READS, int DMAS, int BARRIER, int VALU, int WAITV, int DFORM = 0>
__global__ __launch_bounds__(256) void mix(float* out, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x16)(0.f);
    f32x4 a[2][8];
    for (int i = 0; i < 8; ++i) { a[0][i] = (f32x4)(1e-3f * tid); a[1][i] = (f32x4)(2e-3f * tid); }
    float b = blockIdx.x * 1e-3f + 1.f, vv = tid;
    float w8[8]; int i8[8];
    for (int i = 0; i < 8; ++i) { w8[i] = tid * 0.1f + i; i8[i] = tid * 7 + i; }
    const unsigned s0 = (unsigned)(unsigned long long)(LAS const void*)lds + lane * 16;
    const char* g = (const char*)src + lane * 16 + w * 1024;
    // DFORM 1: SGPR base + 32-bit VGPR offset (no per-piece VALU), M0 from SALU
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const unsigned voff = lane * 16;
    const unsigned long long gb = (unsigned long long)src + wu * 1024;
    const unsigned ub_lo = __builtin_amdgcn_readfirstlane((unsigned)gb), ub_hi = __builtin_amdgcn_readfirstlane((unsigned)(gb >> 32));
    const unsigned long long ub = ((unsigned long long)ub_hi << 32) | ub_lo;
    const unsigned ul = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(LAS const void*)lds + 65536 + wu * 1024);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ub, 0, 1 << 28, 0x00020000);
    __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc((void*)ub, 4, 64, (1 << 23));   // stride 4, 64 records, ADD_TID_ENABLE (DATA_FORMAT bits are stride[17:14] then: keep them 0)
    for (int it = 0; it < iters; ++it) {
        if (WAITV) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
                if constexpr (DMAS > 0 && M % 8 == 0 && M / 8 < DMAS) {
                    if constexpr (DFORM == 0)
                        __builtin_amdgcn_global_load_lds((GAS void*)(g + (M / 8) * 4096), (LAS void*)(lds + 16384 + w * 256 + (M / 8) * 1024), 16, 0, 0);
                    else if constexpr (DFORM == 1)
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ul + (M / 8) * 4096), "v"(voff), "s"(ub + (M / 8) * 4096) : "memory");
                    else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LAS void*)(lds + 16384 + wu * 256 + (M / 8) * 1024), 16, voff, (M / 8) * 4096, 0, 0);
                }
                for (int v = 0; v < VALU; ++v) vv = fmaxf(vv * 1.0001f, 0.5f);
                if constexpr (DFORM == 10 && M % 2 == 0)       // 128 dword stores per body (distinct rows, 128 B apart)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a[gl & 1][i & 7][0]), rs, (int)voff / 4 + (M / 2) * 128, it * 16384, 0);
                if constexpr (DFORM == 13 && M % 2 == 0)       // 128 dword stores, no VGPR address (all lanes one address: issue cost only)
                    asm volatile("buffer_store_dword %0, off, %1, %2 offset:%3" :: "v"(a[gl & 1][i & 7][0]), "s"(rs), "s"(it * 16384), "n"((M / 2) * 16) : "memory");
                if constexpr (DFORM == 14 && M % 2 == 0) {     // 128 dword stores through a descriptor with ADD_TID_ENABLE (lane l -> +4 l bytes)
                    asm volatile("buffer_store_dword %0, off, %1, %2 offset:%3" :: "v"(a[gl & 1][i & 7][0]), "s"(rst), "s"(it * 32768 + (M / 2) * 256), "n"(0) : "memory");
                }
                if constexpr (DFORM == 15 && M % 4 == 0) {     // 64 dwordx2 stores
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    u2 q = {__float_as_uint(a[gl & 1][i & 7][0]), __float_as_uint(a[gl & 1][i & 7][1])};
                    __builtin_amdgcn_raw_buffer_store_b64(q, rs, (int)voff / 2 + (M / 4) * 512, it * 32768, 0);
                }
                if constexpr (DFORM == 11 && M % 8 == 0) {     // 32 dwordx4 stores per body
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    u4 q = {__float_as_uint(a[gl & 1][i & 7][0]), __float_as_uint(a[gl & 1][i & 7][1]), __float_as_uint(a[gl & 1][i & 7][2]), __float_as_uint(a[gl & 1][i & 7][3])};
                    __builtin_amdgcn_raw_buffer_store_b128(q, rs, (int)voff + (M / 8) * 1024, it * 32768, 0);
                }
                if constexpr (DFORM >= 20 && DFORM < 30) {       // (DFORM - 20) independent fma per gap over 8 registers
                    for (int v = 0; v < DFORM - 20; ++v) { const int q = (M * (DFORM - 20) + v) & 7; w8[q] = fmaf(w8[q], 1.0001f, 0.5f); }
                }
                if constexpr (DFORM >= 30 && DFORM < 40) {       // (DFORM - 30) independent integer max per gap (non-idempotent: +1 first)
                    for (int v = 0; v < DFORM - 30; ++v) { const int q = (M * (DFORM - 30) + v) & 7; i8[q] = (i8[q] ^ (int)(M + v)) > 3 ? (i8[q] ^ (int)(M + v)) : 3; }
                }
                if constexpr (DFORM == 12 && M % 2 == 0) {     // 128 independent 1-op VALU (integer max on distinct registers)
                    const int q = (M / 2) & 31;
                    unsigned x = __float_as_uint(a[(gl + 1) & 1][q >> 2 & 7][q & 3]);
                    x = (int)x > 0 ? x : 0u;
                    a[(gl + 1) & 1][q >> 2 & 7][q & 3] = __uint_as_float(x);
                }
                acc[i & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gl & 1][i & 7][i >> 3 & 3], b, acc[i & 15], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    float s = vv;
    for (int i = 0; i < 8; ++i) s += w8[i] + (float)i8[i];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int READS, int DMAS, int BARRIER, int VALU, int WAITV, int DFORM = 0>
void run(const char* name) {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    int grid = p.multiProcessorCount, iters = 2000;
    float *out, *src;
    (void)hipMalloc(&out, grid * 256 * 4); (void)hipMalloc(&src, 1 << 28); (void)hipMemset(src, 0, 1 << 20);
    auto k = mix<READS, DMAS, BARRIER, VALU, WAITV, DFORM>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<grid, 256, 147456>>>(out, src, 200);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<<<grid, 256, 147456>>>(out, src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * 256.0 * 4096.0;
    double cyc = ms * 1e-3 * 2.4e9 / iters;     // cycles per 256-MFMA body at 2.4 GHz (ideal 16384)
    fflush(stdout);
    printf("%-34s %.3f ms  %6.1f TFLOP/s  %7.0f cycles per 256 MFMAs (+%5.0f)\n", name, ms, flop / ms * 1e-9, cyc, cyc - 16384);
    fflush(stdout);
    (void)hipFree(out); (void)hipFree(src);
}

__global__ void tid_probe(float* buf) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 4, 64, (1 << 23));
    float v = 100.f + threadIdx.x;
    asm volatile("buffer_store_dword %0, off, %1, %2 offset:128" :: "v"(v), "s"(r), "s"(1024) : "memory");
}
static void probe() {
    float* d; (void)hipMalloc(&d, 1 << 16); (void)hipMemset(d, 0, 1 << 16);
    tid_probe<<<1, 64>>>(d);
    float h[16384]; (void)hipMemcpy(h, d, 1 << 16, hipMemcpyDeviceToHost);
    int first = -1, cnt = 0, ok = 1;
    for (int i = 0; i < 16384; ++i) if (h[i] != 0.f) { if (first < 0) first = i; ++cnt; }
    for (int l = 0; l < 64; ++l) if (first >= 0 && h[first + l] != 100.f + l) ok = 0;
    printf("add_tid probe: first nonzero float index %d (expect (1024+128)/4 = 288), %d nonzero, contiguous-in-lane-order %d\n", first, cnt, ok);
    fflush(stdout);
}
int main(int argc, char** argv) {
    const int only_new = argc > 1;
    if (only_new) {
        probe();
        run<0, 0, 0, 0, 0>("mfma only (warm-up)");
        run<0, 0, 0, 0, 0>("mfma only");
        run<0, 0, 0, 0, 0, 11>("+ 32 buffer_store_dwordx4");
        run<0, 0, 0, 0, 0, 10>("+ 128 buffer_store_dword");
        run<0, 0, 0, 0, 0, 13>("+ 128 buffer_store_dword off (no vaddr)");
        run<0, 0, 0, 0, 0, 14>("+ 128 buffer_store_dword add_tid");
        run<0, 0, 0, 0, 0, 15>("+ 64 buffer_store_dwordx2");
        return 0;
    }
    run<0, 0, 0, 0, 0>("mfma only");
    run<1, 0, 0, 0, 0>("+ 64 ds_read_b128");
    run<1, 0, 1, 0, 0>("+ reads + barrier");
    run<0, 16, 0, 0, 0>("+ 16 DMA");
    run<0, 8, 0, 0, 0>("+ 8 DMA");
    run<0, 16, 0, 0, 1>("+ 16 DMA + vmcnt(0) per body");
    run<1, 16, 1, 0, 1>("reads + 16 DMA + vmcnt + barrier");
    run<0, 16, 0, 0, 0, 1>("+ 16 DMA (asm saddr form)");
    run<0, 16, 0, 0, 0, 2>("+ 16 DMA (buffer_load lds)");
    run<1, 16, 1, 0, 1, 1>("reads + 16 DMA saddr + vmcnt + bar");
    run<1, 16, 1, 0, 1, 2>("reads + 16 DMA buffer + vmcnt + bar");
    run<0, 0, 0, 0, 0, 10>("+ 128 buffer_store_dword");
    run<0, 0, 0, 0, 0, 11>("+ 32 buffer_store_dwordx4");
    run<0, 0, 0, 0, 0, 12>("+ 128 independent v_max_i32");
    run<0, 0, 0, 1, 0>("+ 1 VALU per gap");
    run<0, 0, 0, 2, 0>("+ 2 VALU per gap");
    run<0, 0, 0, 4, 0>("+ 4 VALU per gap");
    return 0;
}
