// generic.hip -- DM_NeRF for network shapes OTHER than the one the fused kernels are specialised for.
//
// create_nerf (config.py:126-138) passes args.netdepth / args.netwidth / args.multires / args.multires_views through; no
// shipped config changes them (D = 8, W = 256, multires 10 / 4 everywhere), and the register-chained kernels of
// mlp_fwd_impl.h / mlp_bwd.hip / wgrad.hip are built around 256 = 8 accumulator blocks.  So that such a configuration
// RUNS rather than raises, this file provides the layer-by-layer path -- slower (one pass over HBM per layer, 0.31-0.46 of
// the matrix pipe measured), same arithmetic class (f32 MFMA, an fmaf chain over k ascending per output):
//   gemm_kernel       C[i][j] (op)= sum_k A(i,k) B(k,j)  with arbitrary element strides for both operands, so the three
//                     products of a linear layer are one kernel:  forward  Y = X W^T  (+ bias, ReLU),
//                     data gradient  dX = (dY W) . [H > 0],  weight gradient  dW = dY^T X  (split-K over the samples);
//   splitk_reduce_kernel   adds the split-K partials in slice order (deterministic, no float atomics);
//   colsum_kernel     bias gradient, column sums of dY, same two-stage scheme;
//   ray_points_kernel pts = o + d z and the normalised view direction per sample (render.py:37,49-57), feeding
//                     dmnerf_embed; copy_cols_kernel writes an [M, n] block into a column slice (the cat of dm_nerf.py:87,90).
// 128 x 128 block tile, 4 waves each 64 x 64 (2 x 2 v_mfma_f32_32x32x2_f32 tiles), K chunks of 32 staged through LDS
// k-major (every MFMA operand read is one conflict-free ds_read_b32), the next chunk's global loads in flight under the
// current chunk's MFMAs.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dmnerf_hip.h"
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TM = 128, TN = 128, KC = 32;
constexpr int LPT = TM * KC / 256;       // elements per thread, operand and chunk (16)

struct GemmArgs {
    const float* A; int64_t sai, sak;     // A(i, k) = A[i * sai + k * sak]
    const float* B; int64_t sbk, sbj;     // B(k, j) = B[k * sbk + j * sbj]
    float* C; int64_t ldc;                // C[i * ldc + j]   (or the split-K workspace [split][I][J] when splits > 1)
    int64_t I, K; int J;
    const float* bias;                    // [J] or null
    const float* mask; int64_t ldm;       // C *= (mask[i * ldm + j] > 0), or null
    int relu, accumulate, splits;
};

// AK / BK: the operand's unit-stride dimension is k (compile-time: the index arithmetic of the loader folds away)
template <bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs a) {
    __shared__ float As[KC][TM + 1];       // (odd row length: the k-fast operand's stores -- consecutive lanes, consecutive rows -- hit 32 banks)
    __shared__ float Bs[KC][TN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * TM;
    const int j0 = blockIdx.y * TN;
    const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
    // this split's K range
    const int split = blockIdx.z;
    const int64_t kper = ((a.K + a.splits - 1) / a.splits + KC - 1) / KC * KC;
    const int64_t kb = (int64_t)split * kper, ke = kb + kper < a.K ? kb + kper : a.K;
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (f32x16)(0.f);
    // Loader: TM x KC elements per operand and chunk, LPT per thread; the thread index runs along the operand's unit-stride
    // dimension (coalesced 4-byte loads whatever the form: NT forward, NN data gradient, TN weight gradient).  The NEXT
    // chunk is fetched into registers before the current chunk's 64 MFMAs per wave, so the global latency hides under them.
    // Every element slot keeps a 32-bit OFFSET from the tile's (wave-uniform) base pointer -- row / column clamped once, before the
    // loop -- and the base advances by one chunk per trip: a full chunk costs one load per element and no vector ALU (the first
    // version re-derived row, column, both clamps and two 64-bit products per element and chunk, ~400 VALU instructions per
    // thread against the chunk's 64 MFMAs: 0.27-0.37 of the matrix pipe).  Only a K range's partial last chunk is masked.
    constexpr bool a_kfast = AK, b_kfast = BK;
    float ra[LPT], rb[LPT];
    int oa[LPT], ob[LPT];
    const int64_t i_last = a.I - 1 - i0;                               // last valid row / column of this tile, tile-relative
    const int j_last = a.J - 1 - j0;
#pragma unroll
    for (int e = 0; e < LPT; ++e) {
        const int idx = e * 256 + tid;
        const int kk = a_kfast ? (idx & (KC - 1)) : (idx >> 7), ii = a_kfast ? (idx / KC) : (idx & (TM - 1));
        // rows / columns beyond the matrix are read from the clamped (valid) position: they only feed output elements the
        // epilogue never stores
        oa[e] = (int)(((ii < i_last ? ii : i_last) * a.sai + kk * a.sak) * 4);                  // bytes
        const int kb2 = b_kfast ? (idx & (KC - 1)) : (idx >> 7), jj = b_kfast ? (idx / KC) : (idx & (TN - 1));
        ob[e] = (int)((kb2 * a.sbk + (int64_t)(jj < j_last ? jj : j_last) * a.sbj) * 4);
    }
    // MUBUF loads: a descriptor on the tile's (wave-uniform) base + the element's constant 32-bit byte offset in a VGPR; the base
    // moves by SALU arithmetic, so the offsets never leave 32 bits whatever the size of the matrices
    const float* base_a = a.A + i0 * a.sai + kb * a.sak;
    const float* base_b = a.B + kb * a.sbk + (int64_t)j0 * a.sbj;
    const int64_t step_a = (int64_t)KC * a.sak, step_b = (int64_t)KC * a.sbk;
    auto rsrc_of = [](const float* p) {
        const unsigned long long q = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)q), hi = __builtin_amdgcn_readfirstlane((unsigned)(q >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
    };
    auto ld = [](__amdgpu_buffer_rsrc_t r, int byte_off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0)); };
    auto fetch_full = [&]() {
        const __amdgpu_buffer_rsrc_t r_a = rsrc_of(base_a), r_b = rsrc_of(base_b);
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            ra[e] = ld(r_a, oa[e]);
            rb[e] = ld(r_b, ob[e]);
        }
        base_a += step_a;
        base_b += step_b;
    };
    auto fetch_tail = [&](int rem) {                                    // the K range's partial last chunk: rem < KC valid positions
        const __amdgpu_buffer_rsrc_t r_a = rsrc_of(base_a), r_b = rsrc_of(base_b);
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            const int idx = e * 256 + tid;
            const int kk = a_kfast ? (idx & (KC - 1)) : (idx >> 7);
            const int kb2 = b_kfast ? (idx & (KC - 1)) : (idx >> 7);
            // positions beyond the K range: read the last valid one (always inside the matrix), contribute 0
            const float va = ld(r_a, kk < rem ? oa[e] : oa[e] - (int)((kk - rem + 1) * a.sak) * 4);
            const float vb = ld(r_b, kb2 < rem ? ob[e] : ob[e] - (int)((kb2 - rem + 1) * a.sbk) * 4);
            ra[e] = kk < rem ? va : 0.f;
            rb[e] = kb2 < rem ? vb : 0.f;
        }
    };
    auto to_lds = [&]() {
#pragma unroll
        for (int e = 0; e < LPT; ++e) {
            const int idx = e * 256 + tid;
            const int kk = a_kfast ? (idx & (KC - 1)) : (idx >> 7), ii = a_kfast ? (idx / KC) : (idx & (TM - 1));
            As[kk][ii] = ra[e];
            const int kb2 = b_kfast ? (idx & (KC - 1)) : (idx >> 7), jj = b_kfast ? (idx / KC) : (idx & (TN - 1));
            Bs[kb2][jj] = rb[e];
        }
    };
    auto chunk_mfma = [&]() {
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            const int kr = 2 * s + (lane >> 5), c = lane & 31;
            const float a0 = As[kr][wi + c], a1 = As[kr][wi + 32 + c];
            const float b0 = Bs[kr][wj + c], b1 = Bs[kr][wj + 32 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    const int64_t span = ke > kb ? ke - kb : 0;
    const int64_t n_full = span / KC;
    const int rem = (int)(span % KC);
    if (n_full > 0) fetch_full();
    else if (rem) fetch_tail(rem);
    for (int64_t c = 0; c < n_full; ++c) {
        to_lds();
        __syncthreads();
        if (c + 1 < n_full) fetch_full();
        else if (rem) fetch_tail(rem);
        chunk_mfma();
        __syncthreads();
    }
    if (rem) {
        to_lds();
        __syncthreads();
        chunk_mfma();
        __syncthreads();
    }
    // epilogue: lane holds column j = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of each 32 x 32 tile
    float* __restrict__ C = a.splits > 1 ? a.C + (int64_t)split * a.I * a.J : a.C;
    const int64_t ldc = a.splits > 1 ? a.J : a.ldc;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int gj = j0 + wj + 32 * y + (lane & 31);
            if (gj >= a.J) continue;
            const float bj = (a.bias && a.splits == 1) ? a.bias[gj] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gi = i0 + wi + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gi >= a.I) continue;
                float v = acc[x][y][r];
                if (a.splits == 1) {                                    // order: bias, accumulate, ReLU, mask
                    v += bj;
                    if (a.accumulate) v += C[gi * ldc + gj];
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    if (a.mask) v = a.mask[gi * a.ldm + gj] > 0.f ? v : 0.f;
                }
                C[gi * ldc + gj] = v;
            }
        }
}

// out[i * ldo + j] (+)= sum_s part[s][i][j]   (+ bias)   in slice order
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int splits, int64_t I, int J, float* __restrict__ out, int64_t ldo, int accumulate) {
    const int64_t n = I * J;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += part[(int64_t)k * n + e];
        const int64_t i = e / J;
        const int j = (int)(e % J);
        float* o = out + i * ldo + j;
        *o = accumulate ? *o + s : s;
    }
}

// part[slice][j] = sum over the slice's rows of X[m][j]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t ldx, int64_t M, int J, int slices, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int64_t per = (M + slices - 1) / slices, m0 = (int64_t)blockIdx.y * per, m1 = m0 + per < M ? m0 + per : M;
    float s = 0.f;
    if (j < J)
        for (int64_t m = m0 + q; m < m1; m += 4) s += X[m * ldx + j];
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && j < J) part[(int64_t)blockIdx.y * J + j] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void ray_points_kernel(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ z, int64_t N, int S,
                                  float* __restrict__ pts, float* __restrict__ dirs) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= N * S) return;
    const int64_t n = m / S;
    const float dx = rd[n * 3], dy = rd[n * 3 + 1], dz = rd[n * 3 + 2], zv = z[m];
    pts[m * 3 + 0] = ro[n * 3 + 0] + dx * zv;                   // render.py:49: separate multiply and add
    pts[m * 3 + 1] = ro[n * 3 + 1] + dy * zv;
    pts[m * 3 + 2] = ro[n * 3 + 2] + dz * zv;
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);       // render.py:37
    dirs[m * 3 + 0] = dx / nrm; dirs[m * 3 + 1] = dy / nrm; dirs[m * 3 + 2] = dz / nrm;
}

__global__ void copy_cols_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd, int64_t M, int n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * n) return;
    const int64_t m = e / n;
    const int c = (int)(e % n);
    dst[m * ldd + c] = src[m * lds_ + c];
}

}  // namespace

extern "C" int dmnerf_gemm(const float* d_A, int64_t sai, int64_t sak, const float* d_B, int64_t sbk, int64_t sbj, float* d_C, int64_t ldc,
                           int64_t I, int J, int64_t K, const float* d_bias, int relu, const float* d_mask, int64_t ldm, int accumulate,
                           float* d_ws, int splits, void* stream) {
    if (I < 0 || J < 0 || K < 0 || splits < 1) return dmn_fail(DMNERF_E_ARG, "gemm: bad sizes I=%lld J=%d K=%lld splits=%d", (long long)I, J, (long long)K, splits);
    if (I == 0 || J == 0) return DMNERF_OK;
    if (!d_A || !d_B || !d_C) return dmn_fail(DMNERF_E_ARG, "gemm: null pointer");
    if (splits > 1 && (!d_ws || d_bias || relu || d_mask)) return dmn_fail(DMNERF_E_ARG, "gemm: split-K needs a workspace and takes no bias / relu / mask");
    const int64_t ti = (I + TM - 1) / TM;
    const int tj = (J + TN - 1) / TN;
    if (ti > 0x7fffffffLL || tj > 65535 || splits > 65535) return dmn_fail(DMNERF_E_ARG, "gemm: grid too large");
    GemmArgs a{};
    a.A = d_A; a.sai = sai; a.sak = sak; a.B = d_B; a.sbk = sbk; a.sbj = sbj;
    a.C = splits > 1 ? d_ws : d_C; a.ldc = ldc; a.I = I; a.J = J; a.K = K; a.bias = d_bias; a.mask = d_mask; a.ldm = ldm;
    a.relu = relu; a.accumulate = accumulate; a.splits = splits;
    const dim3 grid((unsigned)ti, (unsigned)tj, (unsigned)splits);
    const bool ak = sak == 1, bk = sbk == 1;
    if (ak && bk) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (ak) hipLaunchKernelGGL((gemm_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (bk) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((gemm_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    int rc = dmn_check_launch("gemm");
    if (rc || splits == 1) return rc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (const float*)d_ws, splits, I, J, d_C, ldc, accumulate);
    return dmn_check_launch("gemm: split-K reduce");
}

extern "C" int dmnerf_colsum(const float* d_X, int64_t ldx, int64_t M, int J, float* d_out, float* d_ws, int slices, void* stream) {
    if (M < 0 || J < 1 || slices < 1 || slices > 65535) return dmn_fail(DMNERF_E_ARG, "colsum: bad sizes");
    if (!d_X || !d_out || !d_ws) return dmn_fail(DMNERF_E_ARG, "colsum: null pointer");
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((J + 63) / 64), (unsigned)slices), dim3(256), 0, (hipStream_t)stream, d_X, ldx, M, J, slices, d_ws);
    int rc = dmn_check_launch("colsum");
    if (rc) return rc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, (const float*)d_ws, slices, (int64_t)1, J, d_out, (int64_t)J, 0);
    return dmn_check_launch("colsum: reduce");
}

extern "C" int dmnerf_ray_points(const float* d_rays_o, const float* d_rays_d, const float* d_z, int64_t N, int S, float* d_pts, float* d_dirs, void* stream) {
    if (N < 0 || S < 1) return dmn_fail(DMNERF_E_ARG, "ray_points: bad N=%lld S=%d", (long long)N, S);
    if (N == 0) return DMNERF_OK;
    if (!d_rays_o || !d_rays_d || !d_z || !d_pts || !d_dirs) return dmn_fail(DMNERF_E_ARG, "ray_points: null pointer");
    const int64_t M = N * S;
    hipLaunchKernelGGL(ray_points_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_rays_o, d_rays_d, d_z, N, S, d_pts, d_dirs);
    return dmn_check_launch("ray_points");
}

extern "C" int dmnerf_copy_cols(const float* d_src, int64_t ld_src, float* d_dst, int64_t ld_dst, int64_t M, int n, void* stream) {
    if (M < 0 || n < 0) return dmn_fail(DMNERF_E_ARG, "copy_cols: bad sizes");
    if (M == 0 || n == 0) return DMNERF_OK;
    if (!d_src || !d_dst) return dmn_fail(DMNERF_E_ARG, "copy_cols: null pointer");
    const int64_t total = M * n;
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_src, ld_src, d_dst, ld_dst, M, n);
    return dmn_check_launch("copy_cols");
}
