// wgrad.hip -- weight / bias gradients of the DM-NeRF MLP for gfx950:  dW = dy . x^T  over the batch.
//
// Both operands are the block-major [32-sample block][rows][32] tensors the training forward
// (x: layer inputs) and the dgrad pass (dy: pre-activation gradients) wrote, so every job is an "NT"
// GEMM over K = samples (up to ~10^6) whose output is at most 256 x 256, and the operands of one
// 32-sample chunk are two CONTIGUOUS tiles of rows*128 bytes:
//   * split-K: the samples are cut into slices, one workgroup per (job, slice); slice counts are
//     proportional to the job's FLOPs so that ~one wave of workgroups (<= 256 CUs, one per CU) is
//     balanced; per-slice partials go to a workspace and a second kernel adds them in fixed order
//     (deterministic, no float atomics);
//   * per 32-sample chunk a workgroup stages dy[<=256][32] and x[<=256][32] in LDS through
//     registers (coalesced 128-byte row segments; row stride 34 floats => conflict-free
//     ds_read_b64) with a 2-deep ring, and its 4 waves run v_mfma_f32_32x32x2_f32 with
//     A[i][k] = dy[row i][sample k], B[k][j] = x[row j][sample k];
//   * bias gradients (row sums of dy) ride along in the VALU shadow of wave 0.
// Exact f32 (fmaf-chain MFMA), no vendor BLAS.  Roofline: ~equal parts MFMA (2 x MAC x M FLOP) and
// HBM (each dy / x row is read once per job: ~23 KB per sample and model).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "layout.h"
#include "mlp_common.h"

using namespace dmn;

namespace {

constexpr int KT = 32;            // samples per chunk
constexpr int LDS_STRIDE = 34;    // floats per staged row (32 + 2): banks 34*i mod 64 distinct for i < 32
constexpr int MAX_ROWS = 512;     // A rows + B rows staged per chunk

// One workgroup's work (device table, offsets only => reusable across steps).
struct WgJob {
    int64_t a_off, b_off;     // float offsets of the A / B TENSORS inside their source buffers
    int a_R, b_R;             // total rows of those tensors (block stride = R*32 floats)
    int a_row0, b_row0;       // first row of the job inside the tensor
    int64_t part_off;         // float offset of this workgroup's partial [NBA*32][NBB*32] in the workspace
    int64_t bias_off;         // float offset of its partial row sums [NBA*32], or -1
    int a_src, b_src;         // 0 = saved activations, 1 = dgrad output, 2 = transposed d raw
    int rowsA, rowsB;         // valid rows (the rest of the 32-row blocks is zero)
    int cls;                  // shape class (NBA, NBB)
    int chunk0, nchunk;       // 32-sample chunks [chunk0, chunk0 + nchunk)
    int pad;
};
static_assert(sizeof(WgJob) % 8 == 0, "WgJob layout");

// One output tensor slice (weight columns [col_off, col_off + rowsB) of a parameter, plus its bias).
struct WgOut {
    int64_t part_off, slice_stride;   // first partial, distance between slices
    int64_t bias_part_off, bias_slice_stride;
    int64_t out_off, bias_out_off;    // float offsets into the flat gradient vector (reference order); bias -1 = none
    int n_slices, rowsA, rowsB, ldp;  // ldp = NBB*32
    int ld_out, col_off, pad0, pad1;
};

struct WgArgs {
    const float* src[3];
    float* part;
    const WgJob* jobs;
    int64_t Mp;
    long long* trace;     // diagnostic: per-workgroup {start, end} of the 100 MHz wall clock, or null
};

template <int NBA, int NBB>
struct Split {   // which (A block, B block) pairs a wave owns: a rectangle SA x SB
    static constexpr int SBn = NBB >= 4 ? NBB / 4 : 1;
    static constexpr int SAn = NBB >= 4 ? NBA : (NBB == 2 ? (NBA + 1) / 2 : (NBA + 3) / 4);
    __device__ static int sa(int w, int k) { return NBB >= 4 ? k : (NBB == 2 ? (w >> 1) + 2 * k : w + 4 * k); }
    __device__ static int sb(int w, int k) { return NBB >= 4 ? w + 4 * k : (NBB == 2 ? (w & 1) : 0); }
};

template <int NBA, int NBB>
__device__ __forceinline__ void run_job(const WgArgs& a, const WgJob& jb, float* lds) {
    typedef Split<NBA, NBB> SP;
    constexpr int ROWS = (NBA + NBB) * 32;
    constexpr int NL = NBA + NBB;                  // 16-byte pieces per thread per chunk (ROWS*8/256)
    constexpr int BUF = ROWS * LDS_STRIDE;         // floats per ring slot
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, li = lane & 31;
    // tile of chunk c: rows [row0, row0 + rows) of block (chunk0 + c): contiguous rows*32 floats
    const float* __restrict__ A = a.src[jb.a_src] + jb.a_off + (int64_t)jb.a_row0 * 32;
    const float* __restrict__ B = a.src[jb.b_src] + jb.b_off + (int64_t)jb.b_row0 * 32;
    const int64_t strideA = (int64_t)jb.a_R * 32, strideB = (int64_t)jb.b_R * 32;       // floats per block

    f32x16 acc[SP::SAn * SP::SBn];
#pragma unroll
    for (int i = 0; i < SP::SAn * SP::SBn; ++i) acc[i] = (f32x16)(0.f);
    float bsum[SP::SAn];
#pragma unroll
    for (int i = 0; i < SP::SAn; ++i) bsum[i] = 0.f;
    const bool do_bias = jb.bias_off >= 0 && (NBB >= 4 ? w == 0 : (NBB == 2 ? (w & 1) == 0 : true));

    // per-thread staging geometry: piece e = tid + 256 i -> (row = e >> 3, 16-byte piece = e & 7)
    // pieces i < NBA come from the A tile, the rest from the B tile; inside a tile piece e*4 floats
    const int prow = tid >> 3, pcol = (tid & 7) * 4;
    f32x4 stage[NL];
#ifndef WG_EXP
#define WG_EXP 0
#endif
    auto load_chunk = [&](int c) {
#if WG_EXP == 1
        c &= 1;
#endif
        const float* ta = A + (int64_t)(jb.chunk0 + c) * strideA + tid * 4;
        const float* tb = B + (int64_t)(jb.chunk0 + c) * strideB + tid * 4;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const bool isA = i < NBA;
            const int r = prow + 32 * (isA ? i : i - NBA);
            const bool ok = r < (isA ? jb.rowsA : jb.rowsB);
            const float* p = (isA ? ta : tb) + 1024 * (isA ? i : i - NBA);
            stage[i] = ok ? *reinterpret_cast<const f32x4*>(p) : (f32x4)(0.f);
        }
    };
    auto write_chunk = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            float2* d = reinterpret_cast<float2*>(buf + (prow + 32 * i) * LDS_STRIDE + pcol);   // 8-byte aligned (stride 34, piece*4)
            d[0] = make_float2(stage[i][0], stage[i][1]);
            d[1] = make_float2(stage[i][2], stage[i][3]);
        }
    };

    if (jb.nchunk > 0) {
        load_chunk(0);
        write_chunk(lds);
    }
    __syncthreads();
    for (int c = 0; c < jb.nchunk; ++c) {
        const float* cur = lds + (c & 1) * BUF;
        const bool more = c + 1 < jb.nchunk;
#if WG_EXP != 4
        if (more) load_chunk(c + 1);                    // global loads in flight under the MFMAs below
#endif
        // lane (i = li, kh = half) reads row (blk*32 + i), columns 4t + 2kh, +1 :
        //   MFMA u uses sample 4t + u from lanes 0-31 (k = 0) and sample 4t + 2 + u from lanes 32-63 (k = 1)
        const float* arow[SP::SAn];
        const float* brow[SP::SBn];
#pragma unroll
        for (int k = 0; k < SP::SAn; ++k) arow[k] = cur + (SP::sa(w, k) * 32 + li) * LDS_STRIDE + 2 * half;
#pragma unroll
        for (int k = 0; k < SP::SBn; ++k) brow[k] = cur + ((NBA + SP::sb(w, k)) * 32 + li) * LDS_STRIDE + 2 * half;
        // operands of step t + 1 are read from LDS before the MFMAs of step t issue (register double buffer):
        // with one wave per SIMD nothing else hides the ds_read latency
        float2 av[2][SP::SAn], bv[2][SP::SBn];
        auto read_ops = [&](int t) {
#pragma unroll
            for (int k = 0; k < SP::SAn; ++k) av[t & 1][k] = *reinterpret_cast<const float2*>(arow[k] + 4 * t);
#pragma unroll
            for (int k = 0; k < SP::SBn; ++k) bv[t & 1][k] = *reinterpret_cast<const float2*>(brow[k] + 4 * t);
        };
        read_ops(0);
#pragma unroll
        for (int t = 0; t < KT / 4; ++t) {
            if (t + 1 < KT / 4) read_ops(t + 1);
            // half-way through the chunk the next tile (in flight since the top of the loop) is written to
            // the other ring slot, so the ds_writes issue in the shadow of the remaining MFMAs
#if WG_EXP != 2 && WG_EXP != 4
            if (t == KT / 8 && more) write_chunk(lds + ((c + 1) & 1) * BUF);
#endif
#pragma unroll
            for (int ia = 0; ia < SP::SAn; ++ia) {
                if (NBA * 32 > 0 && SP::sa(w, ia) >= NBA) continue;        // wave has fewer blocks than SAn (wave-uniform)
#pragma unroll
                for (int ib = 0; ib < SP::SBn; ++ib) {
                    acc[ia * SP::SBn + ib] = mfma32(av[t & 1][ia].x, bv[t & 1][ib].x, acc[ia * SP::SBn + ib]);
                    acc[ia * SP::SBn + ib] = mfma32(av[t & 1][ia].y, bv[t & 1][ib].y, acc[ia * SP::SBn + ib]);
                }
                if (do_bias) bsum[ia] += av[t & 1][ia].x + av[t & 1][ia].y;
            }
        }
#if WG_EXP != 3
        __syncthreads();
#endif
    }

    // epilogue: partial tile [NBA*32][NBB*32], C layout: lane holds column j = li, rows crow(r, half)
    float* __restrict__ P = a.part + jb.part_off;
    constexpr int LDP = NBB * 32;
#pragma unroll
    for (int ia = 0; ia < SP::SAn; ++ia) {
        const int ba = SP::sa(w, ia);
        if (ba >= NBA) continue;
#pragma unroll
        for (int ib = 0; ib < SP::SBn; ++ib) {
            const int bb = SP::sb(w, ib);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = ba * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                P[(int64_t)row * LDP + bb * 32 + li] = acc[ia * SP::SBn + ib][r];
            }
        }
        if (do_bias) {
            const float s = bsum[ia] + __shfl_xor(bsum[ia], 32);            // the two k-halves of the row
            if (half == 0) a.part[jb.bias_off + ba * 32 + li] = s;
        }
    }
}

// shape classes (NBA, NBB)
enum { C_8_8 = 0, C_4_8, C_8_2, C_4_1, C_1_8, C_1_4, C_2_4, C_3_4, C_4_4, N_CLASSES };
constexpr int CLS_NBA[N_CLASSES] = {8, 4, 8, 4, 1, 1, 2, 3, 4};
constexpr int CLS_NBB[N_CLASSES] = {8, 8, 2, 1, 8, 4, 4, 4, 4};

__global__ __launch_bounds__(256) void wgrad_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const WgJob jb = a.jobs[blockIdx.x];
    if (a.trace && threadIdx.x == 0) a.trace[2 * blockIdx.x] = (long long)wall_clock64();
    switch (jb.cls) {                                  // workgroup-uniform
        case C_8_8: run_job<8, 8>(a, jb, lds); break;
        case C_4_8: run_job<4, 8>(a, jb, lds); break;
        case C_8_2: run_job<8, 2>(a, jb, lds); break;
        case C_4_1: run_job<4, 1>(a, jb, lds); break;
        case C_1_8: run_job<1, 8>(a, jb, lds); break;
        case C_1_4: run_job<1, 4>(a, jb, lds); break;
        case C_2_4: run_job<2, 4>(a, jb, lds); break;
        case C_3_4: run_job<3, 4>(a, jb, lds); break;
        case C_4_4: run_job<4, 4>(a, jb, lds); break;
        default: break;
    }
    if (a.trace && threadIdx.x == 0) a.trace[2 * blockIdx.x + 1] = (long long)wall_clock64();
}

// Adds the per-slice partials in slice order into the flat gradient vector.
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, const WgOut* __restrict__ outs, int n_outs,
                                    float* __restrict__ grad) {
    const WgOut o = outs[blockIdx.y];
    const int64_t n_w = (int64_t)o.rowsA * o.rowsB;
    const int64_t n_all = n_w + (o.bias_out_off >= 0 ? o.rowsA : 0);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_all; e += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        if (e < n_w) {
            const int r = (int)(e / o.rowsB), c = (int)(e % o.rowsB);
            const float* p = part + o.part_off + (int64_t)r * o.ldp + c;
            for (int k = 0; k < o.n_slices; ++k) s += p[k * o.slice_stride];
            grad[o.out_off + (int64_t)r * o.ld_out + o.col_off + c] = s;
        } else {
            const int r = (int)(e - n_w);
            const float* p = part + o.bias_part_off + r;
            for (int k = 0; k < o.n_slices; ++k) s += p[k * o.bias_slice_stride];
            grad[o.bias_out_off + r] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host: the plan
// ------------------------------------------------------------------------------------------
struct JobDesc {
    int a_src; int64_t a_base; int a_R, a_row0;   // tensor: base (in rows of the SaveLayout), its total rows, first row used
    int b_src; int64_t b_base; int b_R, b_row0;
    int rowsA, rowsB, cls;
    int64_t out_off; int ld_out, col_off;
    int64_t bias_out_off;       // -1: bias handled by another job with the same A
};

int class_for(int rowsA, int rowsB) {
    const int nba = (rowsA + 31) / 32, nbb = (rowsB + 31) / 32;
    for (int c = 0; c < N_CLASSES; ++c)
        if (CLS_NBA[c] == nba && CLS_NBB[c] == nbb) return c;
    return -1;
}

struct Plan {
    std::vector<WgJob> jobs;
    std::vector<WgOut> outs;
    int64_t part_floats = 0;
};

// Parameter offsets in the flat gradient vector = reference state_dict order (weight, bias per layer).
Plan make_plan(int ins_num, int64_t M, int max_wgs) {
    const int C = ins_num + 1;
    const int64_t Mp = save_row_len(M);
    const int nchunks = (int)(Mp / KT);
    // row indices inside the SaveLayout buffers
    const int64_t R_pe = 0, R_de = POS_CH, R_h = POS_CH + DIR_CH, R_f = R_h + 8 * W, R_q = R_f + W, R_g1 = R_q + W, R_g2 = R_g1 + HW;
    // flat gradient offsets
    int64_t off = 0;
    auto lin = [&](int out, int in, int64_t& w_off, int64_t& b_off) { w_off = off; off += (int64_t)out * in; b_off = off; off += out; };
    int64_t w_m[8], b_m[8], w_rf, b_rf, w_if, b_if, w_rh, b_rh, w_ih, b_ih, w_d, b_d, w_io, b_io, w_ro, b_ro;
    lin(W, POS_CH, w_m[0], b_m[0]);
    for (int l = 1; l < 8; ++l) lin(W, l == 5 ? W + POS_CH : W, w_m[l], b_m[l]);
    lin(W, W, w_rf, b_rf); lin(W, W, w_if, b_if); lin(HW, W + DIR_CH, w_rh, b_rh); lin(HW, W, w_ih, b_ih);
    lin(1, W, w_d, b_d); lin(C, HW, w_io, b_io); lin(3, HW, w_ro, b_ro);

    std::vector<JobDesc> d;
    // src ids: 0 = save (x), 1 = dsave (dy), 2 = transposed d raw [blk][4+C][32]
    const int GT = 4 + C;
    d.push_back({1, R_h + 0 * W, W, 0, 0, R_pe, POS_CH, 0, W, POS_CH, 0, w_m[0], POS_CH, 0, b_m[0]});
    for (int l = 1; l < 8; ++l) {
        const int ld = l == 5 ? W + POS_CH : W;
        d.push_back({1, R_h + (int64_t)l * W, W, 0, 0, R_h + (int64_t)(l - 1) * W, W, 0, W, W, 0, w_m[l], ld, 0, b_m[l]});
        if (l == 5) d.push_back({1, R_h + 5 * W, W, 0, 0, R_pe, POS_CH, 0, W, POS_CH, 0, w_m[5], ld, W, -1});   // cat[h, pts] (dm_nerf.py:87)
    }
    d.push_back({1, R_f, W, 0, 0, R_h + 7 * W, W, 0, W, W, 0, w_rf, W, 0, b_rf});
    d.push_back({1, R_q, W, 0, 0, R_h + 7 * W, W, 0, W, W, 0, w_if, W, 0, b_if});
    d.push_back({1, R_g1, HW, 0, 0, R_f, W, 0, HW, W, 0, w_rh, W + DIR_CH, 0, b_rh});
    d.push_back({1, R_g1, HW, 0, 0, R_de, DIR_CH, 0, HW, DIR_CH, 0, w_rh, W + DIR_CH, W, -1});                 // cat[rgb_feature, dirs] (:90)
    d.push_back({1, R_g2, HW, 0, 0, R_q, W, 0, HW, W, 0, w_ih, W, 0, b_ih});
    d.push_back({2, 0, GT, 3, 0, R_h + 7 * W, W, 0, 1, W, 0, w_d, W, 0, b_d});                                  // density_linear
    d.push_back({2, 0, GT, 4, 0, R_g2, HW, 0, C, HW, 0, w_io, HW, 0, b_io});                                   // ins_linear
    d.push_back({2, 0, GT, 0, 0, R_g1, HW, 0, 3, HW, 0, w_ro, HW, 0, b_ro});                                   // rgb_linear
    // Cost of one 32-sample chunk for a workgroup: its MFMA time (NBA*NBB blocks * 16 MFMAs * 64 cycles
    // over 4 SIMDs = 256 cycles per block pair) but never less than the fixed per-chunk latency of the
    // load -> LDS -> barrier pipeline (measured ~2.5 us: the skinny jobs were the long pole when slices
    // were allotted by FLOPs alone).
    auto chunk_cost = [](int cls) { const double c = 256.0 * CLS_NBA[cls] * CLS_NBB[cls]; return c < 6000.0 ? 6000.0 : c; };
    double total = 0;
    for (auto& j : d) { j.cls = class_for(j.rowsA, j.rowsB); total += chunk_cost(j.cls); }

    Plan P;
    for (auto& j : d) {
        const int nba = CLS_NBA[j.cls], nbb = CLS_NBB[j.cls];
        int ns = (int)((double)max_wgs * chunk_cost(j.cls) / total);   // floor => sum <= max_wgs
        if (ns < 1) ns = 1;
        if (ns > nchunks) ns = nchunks;
        const int64_t tile = (int64_t)nba * 32 * nbb * 32, brow = (int64_t)nba * 32;
        WgOut o{};
        o.part_off = P.part_floats; o.slice_stride = tile + brow;
        o.bias_part_off = P.part_floats + tile; o.bias_slice_stride = tile + brow;
        o.out_off = j.out_off; o.bias_out_off = j.bias_out_off;
        o.n_slices = ns; o.rowsA = j.rowsA; o.rowsB = j.rowsB; o.ldp = nbb * 32; o.ld_out = j.ld_out; o.col_off = j.col_off;
        P.outs.push_back(o);
        for (int s = 0; s < ns; ++s) {
            WgJob g{};
            g.a_src = j.a_src; g.b_src = j.b_src;
            g.a_off = j.a_base * Mp; g.b_off = j.b_base * Mp;
            g.a_R = j.a_R; g.b_R = j.b_R; g.a_row0 = j.a_row0; g.b_row0 = j.b_row0;
            g.part_off = P.part_floats; g.bias_off = j.bias_out_off >= 0 ? P.part_floats + tile : -1;
            g.rowsA = j.rowsA; g.rowsB = j.rowsB; g.cls = j.cls;
            g.chunk0 = (int)((int64_t)nchunks * s / ns);
            g.nchunk = (int)((int64_t)nchunks * (s + 1) / ns) - g.chunk0;
            P.jobs.push_back(g);
            P.part_floats += tile + brow;
        }
    }
    return P;
}

}  // namespace

extern "C" int dmnerf_wgrad_plan_sizes(int ins_num, int64_t M, int max_wgs, int64_t* n_job_bytes, int64_t* n_out_bytes,
                                       int64_t* part_floats, int* n_jobs, int* n_outs) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS || M < 1 || max_wgs < 32) return dmn_fail(DMNERF_E_ARG, "wgrad_plan: bad argument");
    const Plan P = make_plan(ins_num, M, max_wgs);
    if (n_job_bytes) *n_job_bytes = (int64_t)(P.jobs.size() * sizeof(WgJob));
    if (n_out_bytes) *n_out_bytes = (int64_t)(P.outs.size() * sizeof(WgOut));
    if (part_floats) *part_floats = P.part_floats;
    if (n_jobs) *n_jobs = (int)P.jobs.size();
    if (n_outs) *n_outs = (int)P.outs.size();
    return DMNERF_OK;
}

extern "C" int dmnerf_wgrad_plan(int ins_num, int64_t M, int max_wgs, void* h_jobs, int64_t job_bytes, void* h_outs, int64_t out_bytes) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS || M < 1 || max_wgs < 32 || !h_jobs || !h_outs) return dmn_fail(DMNERF_E_ARG, "wgrad_plan: bad argument");
    const Plan P = make_plan(ins_num, M, max_wgs);
    if (job_bytes != (int64_t)(P.jobs.size() * sizeof(WgJob)) || out_bytes != (int64_t)(P.outs.size() * sizeof(WgOut)))
        return dmn_fail(DMNERF_E_ARG, "wgrad_plan: buffer sizes do not match dmnerf_wgrad_plan_sizes");
    memcpy(h_jobs, P.jobs.data(), (size_t)job_bytes);
    memcpy(h_outs, P.outs.data(), (size_t)out_bytes);
    return DMNERF_OK;
}

static long long* g_wgrad_trace = nullptr;
extern "C" int dmnerf_wgrad_set_trace(int64_t* d_ticks) {
    g_wgrad_trace = (long long*)d_ticks;
    return DMNERF_OK;
}

extern "C" int dmnerf_mlp_bwd_weights(const float* d_save, const float* d_dsave, const float* d_graw_t, int64_t M,
                                      const void* d_jobs, int n_jobs, const void* d_outs, int n_outs,
                                      float* d_part, float* d_grad_flat, void* stream) {
    if (!d_save || !d_dsave || !d_graw_t || !d_jobs || !d_outs || !d_part || !d_grad_flat || M < 1 || n_jobs < 1 || n_outs < 1)
        return dmn_fail(DMNERF_E_ARG, "mlp_bwd_weights: bad argument");
    WgArgs a{};
    a.src[0] = d_save; a.src[1] = d_dsave; a.src[2] = d_graw_t;
    a.part = d_part; a.jobs = (const WgJob*)d_jobs; a.Mp = save_row_len(M); a.trace = g_wgrad_trace;
    const size_t lds_bytes = 2 * (size_t)MAX_ROWS * LDS_STRIDE * sizeof(float);      // 139 264 B
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return dmn_check_launch("mlp_bwd_weights: hipFuncSetAttribute");
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)n_jobs), dim3(256), lds_bytes, (hipStream_t)stream, a);
    int rc = dmn_check_launch("mlp_bwd_weights");
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(64, (unsigned)n_outs), dim3(256), 0, (hipStream_t)stream,
                       d_part, (const WgOut*)d_outs, n_outs, d_grad_flat);
    return dmn_check_launch("mlp_bwd_weights: reduce");
}
