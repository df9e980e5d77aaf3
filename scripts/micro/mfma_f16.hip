// Microbenchmark + probes (GPU box) for the f16x2 split mode: x ~ hi + lo with hi = f16(x), lo = f16(x - hi),
// a f32 product as THREE v_mfma_f32_32x32x16_f16 (hi.hi + hi.lo + lo.hi).  The scheme lives on f16 SUBNORMAL lo planes
// (weights ~ U(-1/16, 1/16): every lo is below 2^-14), so it needs to be known that
//   1. the operand layout of v_mfma_f32_32x32x16_f16 is the bf16 instruction's,
//   2. the MFMA does NOT flush f16 subnormal inputs,
//   3. v_cvt_pkrtz_f16_f32 produces subnormals and v_fma_mix_f32 reads them (the 2-instruction-per-element split),
//   4. what a one-wave-per-SIMD stream of it sustains with 2 ds_read_b128 per 3 MFMAs (the kernel's operand traffic).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/mfma_f16.hip -o build_exp/mfma_f16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <utility>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define LAS __attribute__((address_space(3)))

template <class F, int... I>
__device__ __forceinline__ void sf_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sf_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- 1 + 2: layout and subnormal inputs.  A [32][16], B [16][32] given as f16 bit patterns
__global__ void probe(const unsigned short* A, const unsigned short* B, float* C) {
    const int l = threadIdx.x, i = l % 32, h = l / 32;
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) {
        a[q] = __builtin_bit_cast(_Float16, A[i * 16 + 8 * h + q]);
        b[q] = __builtin_bit_cast(_Float16, B[(8 * h + q) * 32 + i]);
    }
    f32x16 c = (f32x16)(0.f);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}

// ---- 3: the split as the kernel will issue it: hi = cvt_pkrtz(x0, x1); r = fma_mix(-hi, 1, x); lo = cvt_pkrtz(r0, r1)
__device__ __forceinline__ void split_pair_f16(float x0, float x1, unsigned& whi, unsigned& wlo) {
    float r0, r1;
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %3, %4\n\t"
                 "v_fma_mix_f32 %1, -%0, 1.0, %3 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %2, -%0, 1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(whi), "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1));
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(wlo) : "v"(r0), "v"(r1));
}
__global__ void split_probe(const float* x, unsigned* hi, unsigned* lo, int n2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n2) split_pair_f16(x[2 * i], x[2 * i + 1], hi[i], lo[i]);
}

// ---- 4: stream rate
template <int MODE>   // 0: MFMAs only; 1: + 2 ds_read_b128 per 3 MFMAs (hi group 2 OB MFMAs / OB reads, lo group OB MFMAs / OB reads)
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) lds[i] = 1e-3f * i;
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x16)(0.f);
    f32x4 ah[8], al[8];
    for (int i = 0; i < 8; ++i) { ah[i] = (f32x4)(1e-3f * tid); al[i] = (f32x4)(2e-3f * tid); }
    f32x4 bh = (f32x4)(1.f + blockIdx.x * 1e-3f), bl = (f32x4)(0.5f + blockIdx.x * 1e-3f);
    const unsigned s0 = (unsigned)(unsigned long long)(LAS const void*)lds + lane * 16;
    for (int it = 0; it < iters; ++it) {
        sfor<8>([&](auto kc) {                        // 8 k-blocks x 24 MFMAs = 192 MFMAs per body
            constexpr int kb = decltype(kc)::value;
            if (MODE) { __builtin_amdgcn_s_waitcnt(0xC07F); for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(ah[i])); }
            __builtin_amdgcn_sched_barrier(0);
            sfor<16>([&](auto ic) {                   // hi group: w_hi x (x_hi, x_lo)
                constexpr int i = decltype(ic)::value;
                if constexpr (MODE == 1 && i < 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[i]) : "v"(s0), "n"(((kb * 16 + i) & 63) * 1024));
                acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[i & 7]), __builtin_bit_cast(f16x8, i < 8 ? bh : bl), acc[i & 7], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            if (MODE) { __builtin_amdgcn_s_waitcnt(0xC07F); for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(al[i])); }
            __builtin_amdgcn_sched_barrier(0);
            sfor<8>([&](auto ic) {                    // lo group: w_lo x x_hi; reads the next k-block's hi tiles
                constexpr int i = decltype(ic)::value;
                if constexpr (MODE == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[i]) : "v"(s0), "n"(((kb * 16 + 8 + i) & 63) * 1024));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[i]), __builtin_bit_cast(f16x8, bh), acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name) {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    int grid = p.multiProcessorCount, iters = 3000;
    float* out; (void)hipMalloc(&out, grid * 256 * 4);
    auto k = spin<MODE>;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<grid, 256, 65536>>>(out, 300);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<<<grid, 256, 65536>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * 192.0 * 32768.0;
    printf("%-64s %.3f ms  %7.1f TFLOP/s (dense f16 peak ~2500)\n", name, ms, flop / ms * 1e-9);
    fflush(stdout);
    (void)hipFree(out);
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

int main() {
    // 1. layout (small integers: exact)
    {
        unsigned short hA[512], hB[512]; float fA[512], fB[512], hC[1024], ref[1024];
        for (int i = 0; i < 512; ++i) { fA[i] = (float)((i * 7 + 3) % 13 - 6); fB[i] = (float)((i * 5 + 1) % 11 - 5); hA[i] = f2h(fA[i]); hB[i] = f2h(fB[i]); }
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += fA[i * 16 + k] * fB[k * 32 + j]; ref[i * 32 + j] = s; }
        unsigned short *dA, *dB; float* dC;
        (void)hipMalloc(&dA, 1024); (void)hipMalloc(&dB, 1024); (void)hipMalloc(&dC, 4096);
        (void)hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dC);
        (void)hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 1024; ++i) bad += hC[i] != ref[i];
        printf("v_mfma_f32_32x32x16_f16 layout = the bf16 instruction's: %s (%d of 1024 mismatches)\n", bad ? "WRONG" : "confirmed", bad);
        // 2. subnormal operands: A = subnormal patterns 0x0001 .. (2^-24 k), B = 1.0 / a subnormal B against a normal A
        for (int i = 0; i < 512; ++i) { hA[i] = (unsigned short)(1 + (i % 1023)); hB[i] = f2h(1.0f); }
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += (double)h2f(hA[i * 16 + k]); ref[i * 32 + j] = (float)s; }
        (void)hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dC);
        (void)hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
        bad = 0; int zero = 0; for (int i = 0; i < 1024; ++i) { bad += hC[i] != ref[i]; zero += hC[i] == 0.f; }
        printf("f16 SUBNORMAL A operands x 1.0: %s (%d of 1024 differ from the exact sum, %d are zero; e.g. got %.9g want %.9g)\n",
               bad ? (zero == 1024 ? "FLUSHED TO ZERO" : "INEXACT") : "kept exactly", bad, zero, hC[5], ref[5]);
        for (int i = 0; i < 512; ++i) { hB[i] = (unsigned short)(0x8000 | (1 + (i % 1023))); hA[i] = f2h(2.0f); }
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += 2.0 * (double)h2f(hB[k * 32 + j]); ref[i * 32 + j] = (float)s; }
        (void)hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dC);
        (void)hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
        bad = 0; zero = 0; for (int i = 0; i < 1024; ++i) { bad += hC[i] != ref[i]; zero += hC[i] == 0.f; }
        printf("2.0 x f16 SUBNORMAL (negative) B operands: %s (%d differ, %d zero)\n", bad ? (zero == 1024 ? "FLUSHED TO ZERO" : "INEXACT") : "kept exactly", bad, zero);
        // subnormal x subnormal products (2^-48 range: f32 normal) and accumulation onto a large C are not needed by the scheme
    }
    // 3. split: x = hi + lo + e, |e| <= max(2^-21 |x|, 2^-24)
    {
        const int n = 1 << 16;
        float* hx = new float[n]; unsigned *hhi = new unsigned[n / 2], *hlo = new unsigned[n / 2];
        unsigned s = 12345u;
        for (int i = 0; i < n; ++i) {
            s = s * 1664525u + 1013904223u;
            const float m = 1.0f + (float)(s >> 9) * (1.0f / 8388608.0f);
            const int e = (int)((s >> 3) % 36) - 26;                 // 2^-26 .. 2^9
            hx[i] = ldexpf(m, e) * ((s & 4) ? -1.f : 1.f);
        }
        hx[0] = 0.f; hx[1] = -0.f; hx[2] = 65504.f; hx[3] = 70000.f; hx[4] = 1e-8f; hx[5] = -1e-8f; hx[6] = 6.1e-5f; hx[7] = 0.0625f;
        float* dx; unsigned *dhi, *dlo;
        (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dhi, n * 2); (void)hipMalloc(&dlo, n * 2);
        (void)hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
        split_probe<<<n / 2 / 256, 256>>>(dx, dhi, dlo, n / 2);
        (void)hipMemcpy(hhi, dhi, n * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(hlo, dlo, n * 2, hipMemcpyDeviceToHost);
        int bad = 0, lo_sub = 0, lo_sub_zero = 0; double worst = 0;
        for (int i = 0; i < n; ++i) {
            const unsigned short uh = (unsigned short)(hhi[i / 2] >> (16 * (i & 1))), ul = (unsigned short)(hlo[i / 2] >> (16 * (i & 1)));
            const double hi = h2f(uh), lo = h2f(ul), x = hx[i];
            if (fabs(x) > 65504.0) continue;
            const double e = fabs(x - hi - lo), tol = fmax(ldexp(fabs(x), -21), ldexp(1.0, -24));
            if (fabs(x - hi) >= ldexp(1.0, -14) ? false : (fabs(x - hi) >= ldexp(1.0, -24))) { lo_sub++; lo_sub_zero += (ul & 0x7fff) == 0; }
            if (e > tol || fabs(hi) > fabs(x)) { if (bad < 5) printf("  split off: x %.9g hi %.9g lo %.9g err %.3g tol %.3g\n", x, hi, lo, e, tol); bad++; }
            worst = fmax(worst, e / tol);
        }
        printf("cvt_pkrtz + fma_mix split of %d values in 2^-26..2^9: %d outside max(2^-21 |x|, 2^-24), worst err/tol %.3f; residuals in the f16 subnormal range: %d, "
               "of which flushed to zero by the conversion: %d\n", n, bad, worst, lo_sub, lo_sub_zero);
        const unsigned short u70k = (unsigned short)(hhi[1] >> 16);
        printf("cvt_pkrtz(70000) -> 0x%04x (%g)  [rtz saturates at 65504 instead of inf]\n", u70k, h2f(u70k));
    }
    // 4. stream rate
    run<0>("f16 32x32x16 only, 1 wave/SIMD (warm-up)");
    run<0>("f16 32x32x16 only, 1 wave/SIMD");
    run<1>("+ 2 ds_read_b128 per 3 MFMAs (hi/lo operand traffic), 1 wave/SIMD");
    return 0;
}
