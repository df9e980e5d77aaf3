// Microbenchmark (GPU box): what HBM WRITE rate do the training kernels' activation stores get, and what would other
// shapes of the same traffic get?  The f16x2 training forward writes 7.5 GB of saved activations per 4096 x 192 launch as
// TID-addressed `buffer_store_dword` runs of 256 bytes ("block-major activation tensors"): each of the
// 1024 resident waves walks its own 32 KiB block region of a layer's tensor front to back, 33 runs per ~1.2 us pass, so HBM sees
// ~1000 slow sequential streams interleaved at 256-byte granularity.  With real stores that kernel takes the same WALL time
// whatever the shader clock does (scripts/diag_f16.py EXP_TRAIN=1: 214 k cycles at 1.88 GHz, 174 k at 1.68 GHz, 2.79 ms both
// times; 2.28 ms when the same instructions hit 8 hot blocks), i.e. it is waiting for the memory system at 2.7 TB/s.
//   mode 0  streaming reference: every lane 16 bytes, a workgroup 4 KiB contiguous per instruction, grid-stride
//   mode 1  the kernels' pattern, unpaced: wave w owns blocks w, w + NW, ...; per block and layer 128 runs of 256 B (dword per lane)
//   mode 2  mode 1 with 16-byte lanes (32 runs of 1 KiB per block and layer)
//   mode 3  mode 1 where the four waves of a workgroup interleave their runs in one 128 KiB region (1 KiB contiguous per "row")
//   mode 4  mode 1 paced: PACE dependent 64-cycle v_mfma between stores (the f16x2 training forward asks for ~3.6 TB/s)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/hbm_write.hip -o build_exp/hbm_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ rsrc_t tid_rsrc(float* base) {
    const unsigned long long p = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    float* q = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 4, 64, 1 << 23);          // stride 4 + ADD_TID_ENABLE
}

__global__ __launch_bounds__(256) void stream16(f32x4* out, size_t n16) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = v;
}

// LAYERS tensors of [nblk][256 rows][32 samples] floats; a block region is 32 KiB.  One pass of the real kernel writes one
// out-block quarter (8 KiB = 32 runs) of one layer; order here: layer-major inside a block, like the kernel.
template <int MODE, int PACE>
__global__ __launch_bounds__(256) void rows(float* base, int nblk, int layers, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const size_t layer_floats = (size_t)nblk * 8192;
    f32x16 acc = (f32x16)(0.f);
    const float val = 1.f + lane;
    for (int blk = gw; blk < nblk; blk += nw) {
        for (int L = 0; L < layers; ++L) {
            float* t = base + (size_t)L * layer_floats;
            if (MODE == 1 || MODE == 4) {
                const rsrc_t rs = tid_rsrc(t + (size_t)blk * 8192);
#pragma unroll 8
                for (int r = 0; r < 128; ++r) {
                    if (MODE == 4) {
#pragma unroll
                        for (int p = 0; p < PACE; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(val, val, acc, 0, 0, 0);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rs, 0, r * 256, 0);
                }
            } else if (MODE == 2) {
                f32x4* q = reinterpret_cast<f32x4*>(t + (size_t)blk * 8192);
                const f32x4 v = {val, val, val, val};
#pragma unroll 8
                for (int r = 0; r < 32; ++r) q[r * 64 + lane] = v;
            } else if (MODE == 3) {
                // the workgroup's four blocks as one 128 KiB region; run r of wave w at (4 r + w) * 256
                const int blk0 = blk - wave;            // gw = 4 blockIdx + wave: blk0 is a multiple of 4
                const rsrc_t rs = tid_rsrc(t + (size_t)blk0 * 8192);
#pragma unroll 8
                for (int r = 0; r < 128; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rs, 0, (4 * r + wave) * 256, 0);
            }
        }
    }
    if (MODE == 4 && acc[0] == 123.456f) sink[0] = acc[1];
}

// mode 5: the kernels' instruction mix -- back-to-back INDEPENDENT 32-cycle MFMAs (v_mfma_f32_32x32x16_f16, four accumulators) with
// two 256-byte runs per K MFMAs, PAIRED (both behind the same MFMA, as the SAVE epilogue's phase 1 issues them) or SPREAD (K / 2
// MFMAs apart); HOT: every block folded onto 8 (the stores stay in L2).  MFMA issue alone = runs x K / 2 x 32 cycles.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int K, bool SPREAD, bool HOT>
__global__ __launch_bounds__(256) void mix(float* base, int nblk, int layers, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const size_t layer_floats = (size_t)nblk * 8192;
    f32x16 acc[4] = {(f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f), (f32x16)(0.f)};
    const float val = 1.f + lane;
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(0.001f * (lane + q)); b[q] = (_Float16)(0.002f * (lane - q)); }
    for (int blk = gw; blk < nblk; blk += nw) {
        for (int L = 0; L < layers; ++L) {
            float* t = base + (size_t)L * layer_floats;
            const rsrc_t rs = tid_rsrc(t + (size_t)(HOT ? (blk & 7) : blk) * 8192);
#pragma unroll 4
            for (int r = 0; r < 128; r += 2) {
#pragma unroll
                for (int m = 0; m < K; ++m) {
                    acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
                    if (m == (SPREAD ? K / 2 - 1 : K - 1)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rs, 0, r * 256, 0);
                    if (m == K - 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rs, 0, r * 256 + 256, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.456f) sink[0] = acc[1][1];
}

template <class F>
static float time_ms(F&& launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int nblk = 24576, layers = 9;                           // 24576 blocks of 32 samples = 4096 x 192 samples
    const size_t bytes = (size_t)nblk * 32768 * layers;           // 7.25 GB
    float *buf, *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, bytes));
    const double gb = bytes * 1e-9;
    auto report = [&](const char* name, float ms) { printf("%-72s %7.3f ms  %6.2f TB/s\n", name, ms, gb / ms); };
    report("hipMemsetAsync", time_ms([&] { CK(hipMemsetAsync(buf, 1, bytes, 0)); }, 5));
    for (int g : {1024, 4096, 16384})
        { char n[96]; snprintf(n, 96, "0 streaming, 16 B lanes, %d workgroups", g);
          report(n, time_ms([&] { stream16<<<g, 256>>>((f32x4*)buf, bytes / 16); }, 5)); }
    for (int g : {256, 512, 1024}) {
        char n[96];
        snprintf(n, 96, "1 per-wave 32 KiB regions, 256 B runs, %d workgroups", g);
        report(n, time_ms([&] { rows<1, 0><<<g, 256>>>(buf, nblk, layers, sink); }, 5));
        snprintf(n, 96, "2 per-wave 32 KiB regions, 1 KiB runs, %d workgroups", g);
        report(n, time_ms([&] { rows<2, 0><<<g, 256>>>(buf, nblk, layers, sink); }, 5));
        snprintf(n, 96, "3 workgroup 128 KiB regions, 4 x 256 B adjacent, %d workgroups", g);
        report(n, time_ms([&] { rows<3, 0><<<g, 256>>>(buf, nblk, layers, sink); }, 5));
    }
    // one v_mfma_f32_32x32x2f32 = 64 cycles: at 2.4 GHz PACE p asks for 1024 waves x 256 B / (p x 26.7 ns) = 9.8 / p TB/s
    report("4 pattern 1 paced, 1 MFMA ( 64 cycles) per run: asks 9.8 TB/s", time_ms([&] { rows<4, 1><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    report("4 pattern 1 paced, 2 MFMA (128 cycles) per run: asks 4.9 TB/s", time_ms([&] { rows<4, 2><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    report("4 pattern 1 paced, 3 MFMA (192 cycles) per run: asks 3.3 TB/s", time_ms([&] { rows<4, 3><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    report("4 pattern 1 paced, 4 MFMA (256 cycles) per run: asks 2.5 TB/s", time_ms([&] { rows<4, 4><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    // 27648 runs per wave; MFMA issue alone at 2.4 GHz = 27648 x K / 2 x 32 / 2.4e9
    auto mixrep = [&](const char* what, int K, float ms) {
        char n[128]; const double floor_ms = 27648.0 * K / 2 * 32 / 2.4e6;
        snprintf(n, 128, "5 %s, %d MFMAs per 2 runs (MFMA alone %.3f ms)", what, K, floor_ms);
        report(n, ms);
    };
    mixrep("paired, HBM", 6, time_ms([&] { mix<6, false, false><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    mixrep("spread, HBM", 6, time_ms([&] { mix<6, true, false><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    mixrep("paired, hot", 6, time_ms([&] { mix<6, false, true><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    mixrep("paired, HBM", 8, time_ms([&] { mix<8, false, false><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    mixrep("spread, HBM", 8, time_ms([&] { mix<8, true, false><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    mixrep("paired, hot", 8, time_ms([&] { mix<8, false, true><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    mixrep("paired, HBM", 12, time_ms([&] { mix<12, false, false><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    mixrep("spread, HBM", 12, time_ms([&] { mix<12, true, false><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    mixrep("paired, hot", 12, time_ms([&] { mix<12, false, true><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    mixrep("paired, HBM", 16, time_ms([&] { mix<16, false, false><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    mixrep("paired, hot", 16, time_ms([&] { mix<16, false, true><<<256, 256>>>(buf, nblk, layers, sink); }, 5));
    return 0;
}
