// criterion.hip -- the object-code loss `ins_criterion` (networks/evaluator.py:19-74) on the device.
//
// The reference builds two [labels x channels] cost matrices from the rendered object codes pred [N, C]
// and the per-ray labels (cross-entropy and 1 - soft IoU of every (label, channel) pair, :57-69), copies
// them to the host, lets scipy's linear_sum_assignment match labels to channels (:43-54), and sums the
// matched entries plus the mean prediction of the unmatched channels (:19-37) -- two GPU->CPU syncs per
// training step.  Here the whole loss stays on the stream:
//   cr_partial_kernel   per 64-ray chunk: sums over the chunk's rays of -log(1-P), P (per channel) and of
//                       log(1-P) - log(P), P per (label, channel) -- each element is visited once, its label
//                       picks the LDS row; one thread per channel adds in ray order (deterministic)
//   cr_reduce_kernel    the chunk partials of every entry summed in f64 (16 lanes per entry, fixed order), many workgroups:
//                       with one workgroup this latency-bound sum was 0.8 of the 1.0 ms the loss took at ins_num 93
//   cr_solve_kernel     sums -> cost matrices (rows = the labels that occur, ascending: the one-hot
//                       compaction of :21-26), then the rectangular assignment by shortest augmenting paths
//                       (the algorithm scipy implements, Crouse 2016) on one wavefront with the column scan
//                       spread over the lanes, then the three loss terms
//   cr_bwd_kernel       d loss / d pred, elementwise (matched channels: cross-entropy + soft-IoU terms,
//                       unmatched channels: 1 / (N U))
// Sums are accumulated in f32 per 64-ray chunk and in f64 across chunks; the cost entries are rounded to f32
// and added in f32 like the reference's `cost_ce + cost_siou` before the solver sees them as doubles.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dmnerf_hip.h"
#include "common.h"


namespace {

constexpr int CR_CHUNK = 64;      // rays per partial-sum block
constexpr int CR_MAXC = 128;      // channels (= ins_num) supported by the solver's lane mapping (2 columns per lane)

// Work buffer layout (byte offsets; everything 8-byte aligned).  L = C + 1 label values 0..C.
struct CrLayout {
    int64_t part_b, part_t, part_a, part_s, part_cnt;     // chunk partials
    int64_t part_flag;                                    // int [nch]: DMNERF_CRIT_* conditions seen by the chunk (plain stores: nothing to zero)
    int64_t red;                                          // double [L + 2 C + 2 L C]: counts | A | S | b | t summed over the chunks
    int64_t ce, siou, tp_all;                             // float [C][C]: rows g < V (tp_all: the soft true-positive sums)
    int64_t row4col, lab_of_row, tp_of_col, den_of_col;   // int [C], int [C], float [C], float [C]
    int64_t scal;                                         // int V, int U, int flags (DMNERF_CRIT_*), pad
    int64_t total;
    int nch, L;
};

__host__ __device__ inline CrLayout cr_layout(int64_t N, int C) {
    CrLayout w{};
    w.nch = (int)((N + CR_CHUNK - 1) / CR_CHUNK);
    w.L = C + 1;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { const int64_t at = o; o += (bytes + 7) & ~(int64_t)7; return at; };
    w.part_b = take((int64_t)w.nch * w.L * C * 4);
    w.part_t = take((int64_t)w.nch * w.L * C * 4);
    w.part_a = take((int64_t)w.nch * C * 4);
    w.part_s = take((int64_t)w.nch * C * 4);
    w.part_cnt = take((int64_t)w.nch * w.L * 4);
    w.part_flag = take((int64_t)w.nch * 4);
    w.red = take((int64_t)(w.L + 2 * C + 2 * w.L * C) * 8);
    w.ce = take((int64_t)C * C * 4);
    w.siou = take((int64_t)C * C * 4);
    w.tp_all = take((int64_t)C * C * 4);
    w.row4col = take((int64_t)C * 4);
    w.lab_of_row = take((int64_t)C * 4);
    w.tp_of_col = take((int64_t)C * 4);
    w.den_of_col = take((int64_t)C * 4);
    w.scal = take(16);
    w.total = o;
    return w;
}

// One or two problems per launch (blockIdx.y): the two levels of a training step -- pred_fine / pred_coarse against the SAME
// labels, N and C -- go through every kernel together (dmnerf_ins_criterion_fwd2 / _bwd2: 3 + 1 launches per step instead of 8 + 2).
struct CrPair {
    const float* pred[2];
    char* work[2];
    float* out4[2];
    const float* gout4[2];
    float* grad[2];
};

// ---- per-chunk partial sums ---------------------------------------------------------------------------
__global__ __launch_bounds__(CR_MAXC) void cr_partial_kernel(const CrPair pr, const int* __restrict__ labels, int64_t N, int C) {
    extern __shared__ float lds[];                       // [L][C] b-sums, [L][C] P-sums, [L] counts
    const float* __restrict__ pred = pr.pred[blockIdx.y];
    char* __restrict__ work = pr.work[blockIdx.y];
    const CrLayout w = cr_layout(N, C);
    const int L = w.L, p = threadIdx.x;
    float* lb = lds;
    float* lt = lds + L * C;
    int* lc = reinterpret_cast<int*>(lds + 2 * L * C);
    for (int i = p; i < 2 * L * C + L; i += blockDim.x) lds[i] = 0.f;       // (int 0 == float 0 bit pattern)
    __syncthreads();
    const int64_t n0 = (int64_t)blockIdx.x * CR_CHUNK;
    float acc_a = 0.f, acc_s = 0.f;
    int bad = 0;
    for (int r = 0; r < CR_CHUNK; ++r) {
        const int64_t n = n0 + r;
        if (n >= N) break;
        const int l = labels[n];
        const bool lab_ok = l >= 0 && l < L;
        if (p < C) {
            const float P = pred[n * C + p];
            const float x = logf(P + 1e-8f);                 // log(pred + 1e-8)             (:57)
            const float y = logf((1.f - P) + 1e-8f);         // log(1 - pred + 1e-8)
            acc_a -= y;
            acc_s += P;
            if (lab_ok) {                                    // this thread owns column p of every LDS row: no race
                lb[l * C + p] += (float)((double)y - (double)x);
                lt[l * C + p] += P;
            }
        }
        if (p == 0 && lab_ok) lc[l] += 1;
        // a label outside [0, ins_num]: the reference's one_hot / column indexing (:21-25) would raise; here the ray
        // takes part in no row, and the condition is reported in the flags word (through this chunk's flag, which cr_solve_kernel ORs)
        if (!lab_ok) bad = DMNERF_CRIT_LABEL_RANGE;
    }
    __syncthreads();
    float* pb = reinterpret_cast<float*>(work + w.part_b) + (int64_t)blockIdx.x * L * C;
    float* pt = reinterpret_cast<float*>(work + w.part_t) + (int64_t)blockIdx.x * L * C;
    for (int i = p; i < L * C; i += blockDim.x) { pb[i] = lb[i]; pt[i] = lt[i]; }
    if (p < C) {
        reinterpret_cast<float*>(work + w.part_a)[(int64_t)blockIdx.x * C + p] = acc_a;
        reinterpret_cast<float*>(work + w.part_s)[(int64_t)blockIdx.x * C + p] = acc_s;
    }
    int* pc = reinterpret_cast<int*>(work + w.part_cnt) + (int64_t)blockIdx.x * L;
    for (int i = p; i < L; i += blockDim.x) pc[i] = lc[i];
    if (p == 0) reinterpret_cast<int*>(work + w.part_flag)[blockIdx.x] = bad;
}

// ---- sums over the chunks: entry e of [counts (L) | A (C) | S (C) | b (L C) | t (L C)], 16 neighbouring lanes per entry (every 16th
// chunk partial each, loads in flight together), combined in a fixed butterfly order; exact for the counts
__global__ __launch_bounds__(256) void cr_reduce_kernel(int64_t N, int C, const CrPair pr) {
    char* __restrict__ work = pr.work[blockIdx.y];
    const CrLayout w = cr_layout(N, C);
    const int L = w.L, sub = threadIdx.x & 15;
    const int e = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int n_ent = L + 2 * C + 2 * L * C;
    double v = 0.0;
    if (e < L) {
        const int* __restrict__ pc = reinterpret_cast<const int*>(work + w.part_cnt);
        int c = 0;
        for (int k = sub; k < w.nch; k += 16) c += pc[(int64_t)k * L + e];
        v = (double)c;
    } else if (e < n_ent) {
        const float* __restrict__ src;
        int64_t stride;
        int off;
        if (e < L + C) { src = reinterpret_cast<const float*>(work + w.part_a); stride = C; off = e - L; }
        else if (e < L + 2 * C) { src = reinterpret_cast<const float*>(work + w.part_s); stride = C; off = e - L - C; }
        else if (e < L + 2 * C + L * C) { src = reinterpret_cast<const float*>(work + w.part_b); stride = (int64_t)L * C; off = e - L - 2 * C; }
        else { src = reinterpret_cast<const float*>(work + w.part_t); stride = (int64_t)L * C; off = e - L - 2 * C - L * C; }
        for (int k = sub; k < w.nch; k += 16) v += (double)src[(int64_t)k * stride + off];
    }
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    if (sub == 0 && e < n_ent) reinterpret_cast<double*>(work + w.red)[e] = v;
}

// ---- cost matrices, assignment, loss terms: one workgroup ------------------------------------------------
__device__ __forceinline__ double wave_min_key(double v, int key, int& key_out) {
    // minimum of v over the wave; ties: smaller key.  Returns the minimum, key_out = its key.
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int ok = __shfl_xor(key, off);
        if (ov < v || (ov == v && ok < key)) { v = ov; key = ok; }
    }
    key_out = key;
    return v;
}

__global__ __launch_bounds__(256) void cr_solve_kernel(int64_t N, int C, const CrPair pr) {
    char* __restrict__ work = pr.work[blockIdx.x];
    float* __restrict__ out4 = pr.out4[blockIdx.x];
    const CrLayout w = cr_layout(N, C);
    const int L = w.L, tid = threadIdx.x;
    __shared__ int s_cnt[CR_MAXC + 1], s_rank[CR_MAXC + 1], s_V;
    __shared__ double s_A[CR_MAXC], s_S[CR_MAXC];
    __shared__ double s_u[CR_MAXC], s_v[CR_MAXC], s_spc[CR_MAXC];
    __shared__ int s_path[CR_MAXC], s_col4row[CR_MAXC], s_row4col[CR_MAXC];
    __shared__ unsigned char s_SR[CR_MAXC], s_SC[CR_MAXC];
    __shared__ int s_flags;
    if (tid == 0) s_flags = 0;
    __syncthreads();
    {   // the chunks' condition flags -> one word (this kernel writes V, U and the flags word afresh on every call)
        const int* pf = reinterpret_cast<const int*>(work + w.part_flag);
        int f = 0;
        for (int k = tid; k < w.nch; k += blockDim.x) f |= pf[k];
        if (f) atomicOr(&s_flags, f);
    }
    float* ce = reinterpret_cast<float*>(work + w.ce);
    float* siou = reinterpret_cast<float*>(work + w.siou);
    float* tp_all = reinterpret_cast<float*>(work + w.tp_all);
    const double* red = reinterpret_cast<const double*>(work + w.red);      // cr_reduce_kernel: counts | A | S | b | t
    const double* red_b = red + L + 2 * C;
    const double* red_t = red_b + L * C;

    // 1. label counts, the labels that occur (ascending) -> rows (evaluator.py:21-26)
    for (int e = tid; e < L + 2 * C; e += blockDim.x) {
        const double v = red[e];
        if (e < L) s_cnt[e] = (int)v;
        else if (e < L + C) s_A[e - L] = v;
        else s_S[e - L - C] = v;
    }
    __syncthreads();
    if (tid == 0) {
        int v = 0;
        int extra = 0;
        for (int l = 0; l < L; ++l) {
            if (s_cnt[l] > 0 && v >= C) ++extra;          // more distinct labels than channels: the reference raises on
            s_rank[l] = (s_cnt[l] > 0 && v < C) ? v++ : -1;   // the one-hot column mismatch (:24); here the first C are kept
        }
        if (extra) atomicOr(&s_flags, DMNERF_CRIT_TOO_MANY_LABELS);
        s_V = v;
    }
    __syncthreads();
    const int V = s_V;
    int* lab_of_row = reinterpret_cast<int*>(work + w.lab_of_row);
    for (int l = tid; l < L; l += blockDim.x)
        if (s_rank[l] >= 0) lab_of_row[s_rank[l]] = l;

    // 2. cost matrices (evaluator.py:57-66), f32 entries
    for (int e = tid; e < L * C; e += blockDim.x) {
        const int l = e / C, p = e - l * C;
        const int g = s_rank[l];
        if (g >= 0) {
            ce[g * C + p] = (float)((s_A[p] + red_b[e]) / (double)N);
            const float TP = (float)red_t[e];
            tp_all[g * C + p] = TP;
            const float FP = (float)s_S[p] - TP;
            const float FN = (float)s_cnt[l] - TP;
            siou[g * C + p] = 1.0f - TP / (TP + FP + FN + 1e-6f);
        }
    }
    __syncthreads();

    // 3. rectangular assignment of the V label rows to the C channels (V <= C) by shortest augmenting paths
    //    (evaluator.py:43-47 -> scipy linear_sum_assignment).  Wave 0; lane j owns columns j and j + 64.
    if (tid < 64) {
        const int lane = tid;
        for (int j = lane; j < C; j += 64) { s_v[j] = 0.0; s_row4col[j] = -1; }
        for (int i = lane; i < V; i += 64) { s_u[i] = 0.0; s_col4row[i] = -1; }
        __builtin_amdgcn_wave_barrier();
        for (int cur = 0; cur < V; ++cur) {
            for (int j = lane; j < C; j += 64) { s_spc[j] = __builtin_inf(); s_SC[j] = 0; s_path[j] = -1; }
            for (int i = lane; i < V; i += 64) s_SR[i] = 0;
            __builtin_amdgcn_wave_barrier();
            double min_val = 0.0;
            int i = cur, sink = -1;
            while (sink < 0) {
                if (lane == 0) s_SR[i] = 1;
                const double ui = s_u[i];
                double best = __builtin_inf();
                int best_key = 0x7fffffff;
                for (int j = lane; j < C; j += 64) {
                    if (s_SC[j]) continue;
                    const double cost = (double)(ce[i * C + j] + siou[i * C + j]);      // f32 add as in `cost_ce + cost_siou` (:69)
                    const double r = min_val + cost - ui - s_v[j];
                    if (r < s_spc[j]) { s_spc[j] = r; s_path[j] = i; }
                    const double sj = s_spc[j];
                    // ties: an unassigned column first (it ends the search), then the lower index
                    const int key = (s_row4col[j] < 0 ? 0 : CR_MAXC) + j;
                    if (sj < best || (sj == best && key < best_key)) { best = sj; best_key = key; }
                }
                int key;
                min_val = wave_min_key(best, best_key, key);
                const int jstar = key >= CR_MAXC ? key - CR_MAXC : key;
                if (lane == 0) s_SC[jstar] = 1;
                __builtin_amdgcn_wave_barrier();
                if (s_row4col[jstar] < 0) sink = jstar; else i = s_row4col[jstar];
            }
            // dual updates
            for (int r = lane; r < V; r += 64)
                if (s_SR[r]) s_u[r] += (r == cur) ? min_val : min_val - s_spc[s_col4row[r]];
            for (int j = lane; j < C; j += 64)
                if (s_SC[j]) s_v[j] -= min_val - s_spc[j];
            __builtin_amdgcn_wave_barrier();
            // augment along the path (serial; at most V steps)
            if (lane == 0) {
                int j = sink;
                while (true) {
                    const int r = s_path[j];
                    s_row4col[j] = r;
                    const int prev = s_col4row[r];
                    s_col4row[r] = j;
                    j = prev;
                    if (r == cur) break;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    // 4. loss terms (evaluator.py:28-37) and what the backward needs
    int* row4col = reinterpret_cast<int*>(work + w.row4col);
    float* tp_of_col = reinterpret_cast<float*>(work + w.tp_of_col);
    float* den_of_col = reinterpret_cast<float*>(work + w.den_of_col);
    for (int p = tid; p < C; p += blockDim.x) {
        const int g = s_row4col[p];
        row4col[p] = g;
        if (g >= 0) {
            const int l = lab_of_row[g];
            const float TP = tp_all[g * C + p];                          // the value the cost matrix used
            tp_of_col[p] = TP;
            den_of_col[p] = TP + ((float)s_S[p] - TP) + ((float)s_cnt[l] - TP) + 1e-6f;
        }
    }
    if (tid == 0) {
        double sce = 0.0, ssi = 0.0, sinv = 0.0;
        for (int g = 0; g < V; ++g) { sce += (double)ce[g * C + s_col4row[g]]; ssi += (double)siou[g * C + s_col4row[g]]; }
        int U = 0;
        for (int p = 0; p < C; ++p)
            if (s_row4col[p] < 0) { sinv += s_S[p]; ++U; }
        const float valid_ce = V > 0 ? (float)(sce / V) : 0.f;
        const float valid_siou = V > 0 ? (float)(ssi / V) : 0.f;
        const float invalid_ce = U > 0 ? (float)(sinv / ((double)N * U)) : 0.f;     // torch.tensor([0]) when every channel is matched (:33)
        out4[0] = valid_ce + invalid_ce + valid_siou;
        out4[1] = valid_ce; out4[2] = invalid_ce; out4[3] = valid_siou;
        int* sc = reinterpret_cast<int*>(work + w.scal);
        sc[0] = V; sc[1] = U; sc[2] = s_flags; sc[3] = 0;
    }
}

// ---- backward: elementwise -------------------------------------------------------------------------------
__global__ void cr_bwd_kernel(const CrPair pr, const int* __restrict__ labels, int64_t N, int C) {
    const float* __restrict__ pred = pr.pred[blockIdx.y];
    const char* __restrict__ work = pr.work[blockIdx.y];
    const float* __restrict__ gout4 = pr.gout4[blockIdx.y];
    float* __restrict__ grad = pr.grad[blockIdx.y];
    const CrLayout w = cr_layout(N, C);
    const int* row4col = reinterpret_cast<const int*>(work + w.row4col);
    const int* lab_of_row = reinterpret_cast<const int*>(work + w.lab_of_row);
    const float* tp_of_col = reinterpret_cast<const float*>(work + w.tp_of_col);
    const float* den_of_col = reinterpret_cast<const float*>(work + w.den_of_col);
    const int* sc = reinterpret_cast<const int*>(work + w.scal);
    const int V = sc[0], U = sc[1];
    // ins_loss_sum = valid_ce + invalid_ce + valid_siou, each of the four is also an output
    const float c_ce = gout4[0] + gout4[1], c_inv = gout4[0] + gout4[2], c_si = gout4[0] + gout4[3];
    const int64_t total = N * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = e / C;
        const int p = (int)(e - n * C);
        const int g = row4col[p];
        float d = 0.f;
        if (g >= 0) {
            const float P = pred[e];
            const bool G = labels[n] == lab_of_row[g];
            // d/dP of mean_n(-G log(P + 1e-8) - (1 - G) log(1 - P + 1e-8)), then the mean over the V matched pairs
            const float dce = (G ? -1.f / (P + 1e-8f) : 1.f / ((1.f - P) + 1e-8f)) / ((float)N * (float)V);
            // cost_siou = 1 - TP / D, D = TP + FP + FN + 1e-6:  dTP/dP = G, dD/dP = 1 - G
            const float TP = tp_of_col[p], D = den_of_col[p];
            const float dsi = -((G ? D : 0.f) - (G ? 0.f : TP)) / (D * D) / (float)V;
            d = c_ce * dce + c_si * dsi;
        } else if (U > 0) {
            d = c_inv / ((float)N * (float)U);
        }
        grad[e] = d;
    }
}

}  // namespace

extern "C" int64_t dmnerf_ins_criterion_work_bytes(int64_t N, int ins_num) {
    if (N < 1 || ins_num < 1 || ins_num > CR_MAXC) return -1;
    return cr_layout(N, ins_num).total;
}

extern "C" int64_t dmnerf_ins_criterion_flags_offset(int64_t N, int ins_num) {
    if (N < 1 || ins_num < 1 || ins_num > CR_MAXC) return -1;
    return cr_layout(N, ins_num).scal + 8;
}

static int cr_forward(const CrPair& pr, int levels, const int32_t* d_labels, int64_t N, int ins_num, int64_t work_bytes, void* stream) {
    if (N < 1 || ins_num < 1 || ins_num > CR_MAXC) return dmn_fail(DMNERF_E_ARG, "ins_criterion: bad N=%lld ins_num=%d (max %d)", (long long)N, ins_num, CR_MAXC);
    for (int l = 0; l < levels; ++l)
        if (!pr.pred[l] || !pr.work[l] || !pr.out4[l]) return dmn_fail(DMNERF_E_ARG, "ins_criterion: null pointer");
    if (!d_labels) return dmn_fail(DMNERF_E_ARG, "ins_criterion: null pointer");
    const CrLayout w = cr_layout(N, ins_num);
    if (work_bytes < w.total) return dmn_fail(DMNERF_E_ARG, "ins_criterion: work buffer too small (%lld < %lld bytes)", (long long)work_bytes, (long long)w.total);
    const size_t lds = (size_t)(2 * w.L * ins_num + w.L) * sizeof(float);
    static DmnOncePerDevice once;
    if (hipError_t e = once.run([] { return hipFuncSetAttribute((const void*)cr_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                2 * (CR_MAXC + 1) * CR_MAXC * 4 + (CR_MAXC + 1) * 4); });
        e != hipSuccess)
        return dmn_fail_hip(e, "ins_criterion: hipFuncSetAttribute");
    // (no memset: V, U and the flags word are written afresh by cr_solve_kernel; the chunks report through plain stores)
    hipLaunchKernelGGL(cr_partial_kernel, dim3((unsigned)w.nch, (unsigned)levels), dim3(CR_MAXC), lds, (hipStream_t)stream, pr, (const int*)d_labels, N, ins_num);
    int rc = dmn_check_launch("ins_criterion: partial sums");
    if (rc) return rc;
    const int n_ent = w.L + 2 * ins_num + 2 * w.L * ins_num;
    hipLaunchKernelGGL(cr_reduce_kernel, dim3((unsigned)((n_ent + 15) / 16), (unsigned)levels), dim3(256), 0, (hipStream_t)stream, N, ins_num, pr);
    rc = dmn_check_launch("ins_criterion: chunk sums");
    if (rc) return rc;
    hipLaunchKernelGGL(cr_solve_kernel, dim3((unsigned)levels), dim3(256), 0, (hipStream_t)stream, N, ins_num, pr);
    return dmn_check_launch("ins_criterion: solve");
}

static int cr_backward(const CrPair& pr, int levels, const int32_t* d_labels, int64_t N, int ins_num, void* stream) {
    if (N < 1 || ins_num < 1 || ins_num > CR_MAXC) return dmn_fail(DMNERF_E_ARG, "ins_criterion_bwd: bad N=%lld ins_num=%d", (long long)N, ins_num);
    for (int l = 0; l < levels; ++l)
        if (!pr.pred[l] || !pr.work[l] || !pr.gout4[l] || !pr.grad[l]) return dmn_fail(DMNERF_E_ARG, "ins_criterion_bwd: null pointer");
    if (!d_labels) return dmn_fail(DMNERF_E_ARG, "ins_criterion_bwd: null pointer");
    const int64_t total = N * ins_num;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(cr_bwd_kernel, dim3(blocks, (unsigned)levels), dim3(256), 0, (hipStream_t)stream, pr, (const int*)d_labels, N, ins_num);
    return dmn_check_launch("ins_criterion_bwd");
}

extern "C" int dmnerf_ins_criterion_fwd(const float* d_pred, const int32_t* d_labels, int64_t N, int ins_num, void* d_work,
                                        int64_t work_bytes, float* d_out4, void* stream) {
    CrPair pr{};
    pr.pred[0] = d_pred; pr.work[0] = (char*)d_work; pr.out4[0] = d_out4;
    return cr_forward(pr, 1, d_labels, N, ins_num, work_bytes, stream);
}

extern "C" int dmnerf_ins_criterion_bwd(const float* d_pred, const int32_t* d_labels, int64_t N, int ins_num, const void* d_work,
                                        const float* d_gout4, float* d_grad_pred, void* stream) {
    CrPair pr{};
    pr.pred[0] = d_pred; pr.work[0] = (char*)const_cast<void*>(d_work); pr.gout4[0] = d_gout4; pr.grad[0] = d_grad_pred;
    return cr_backward(pr, 1, d_labels, N, ins_num, stream);
}

extern "C" int dmnerf_ins_criterion_fwd2(const float* d_pred_a, const float* d_pred_b, const int32_t* d_labels, int64_t N, int ins_num,
                                         void* d_work_a, void* d_work_b, int64_t work_bytes, float* d_out4_a, float* d_out4_b, void* stream) {
    CrPair pr{};
    pr.pred[0] = d_pred_a; pr.pred[1] = d_pred_b; pr.work[0] = (char*)d_work_a; pr.work[1] = (char*)d_work_b;
    pr.out4[0] = d_out4_a; pr.out4[1] = d_out4_b;
    return cr_forward(pr, 2, d_labels, N, ins_num, work_bytes, stream);
}

extern "C" int dmnerf_ins_criterion_bwd2(const float* d_pred_a, const float* d_pred_b, const int32_t* d_labels, int64_t N, int ins_num,
                                         const void* d_work_a, const void* d_work_b, const float* d_gout4_a, const float* d_gout4_b,
                                         float* d_grad_a, float* d_grad_b, void* stream) {
    CrPair pr{};
    pr.pred[0] = d_pred_a; pr.pred[1] = d_pred_b;
    pr.work[0] = (char*)const_cast<void*>(d_work_a); pr.work[1] = (char*)const_cast<void*>(d_work_b);
    pr.gout4[0] = d_gout4_a; pr.gout4[1] = d_gout4_b; pr.grad[0] = d_grad_a; pr.grad[1] = d_grad_b;
    return cr_backward(pr, 2, d_labels, N, ins_num, stream);
}
