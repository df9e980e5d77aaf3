// gemm_tn.hip -- the WEIGHT gradient of a linear layer of network shapes other than the shipped one (config.py:31-41 netdepth / netwidth /
// multires* -> create_nerf :126-138), with its bias gradient:  dW [n_out][n_in] = dy^T x,  db [n_out] = column sums of dy  over the M
// samples -- the "TN" product: BOTH operands are sample-major rows ([M][ld], what gemm_nt.hip reads and writes), and the reduction runs
// over the rows.  (csrc/generic.hip's strided kernel did it with per-lane global loads and a separate column-sum launch: 250 .. 500 us
// + 310 us + a 50 us reduction per layer at 786 432 samples, a quarter of a training step of a W = 128 network.)
//
//   * a chunk = 32 samples: the tiles dy[32][32 TA] and x[32][32 TB] go global -> LDS by LDS-DMA (a 1-KiB piece = 8 samples x 128 B of
//     one 32-column block; wave w fetches piece w of every block) into a D-deep ring -- row-major, no swizzle: the reads below are
//     4 bytes wide and a lane half reads one whole 128-byte row segment;
//   * v_mfma_f32_32x32x2_f32 with A[i][k] = dy[sample k][column i], B[k][j] = x[sample k][column j]: lane (li, half) reads column li of
//     sample 2 s + half with ONE ds_read_b32 per operand block and k-step (both operands are "column per lane" here -- no 16-byte read
//     serves four k-steps as in the NT kernels); a wave owns SA x SB blocks of the TA x TB tile (waves WA x WB), so a k-step is SA + SB
//     reads for SA SB MFMAs;
//   * the reads of chunk c + 1 go into a second register set while the MFMAs of chunk c run (hand-over per chunk: vmcnt for the ring,
//     lgkmcnt(0) for the registers, one barrier); the refill of the slot chunk c just left follows the barrier;
//   * the bias gradient rides along: the A registers ARE dy's columns -- one v_add_f32 per A read (the waves of column wb = 0);
//   * split-K: one workgroup per (output tile, slice of the samples), slices = CUs / tiles; per-slice partial tiles and column sums go
//     to a workspace and reduce_tn_kernel adds them in slice order (deterministic: no float atomics).
// Exact f32.  Roofline: HBM and MFMA in about equal parts for a W x W layer (each dy / x row is read once per output tile column / row:
// 8 W bytes per sample against 2 W^2 flops), HBM for the skinny ones (n_out = 1, 3, C; n_in = 27, 63).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "mlp_common.h"

using namespace dmn;

namespace {

constexpr int TN_LDS_BUDGET = 163840;

struct TnArgs {
    const float* A; int64_t lda, a_floats;      // dy [M][lda] (floats from the pointer to the end of its allocation)
    const float* B; int64_t ldb, b_floats;      // x  [M][ldb]
    int64_t M, chunks;                          // chunks = ceil(M / 32)
    int tiles_b, slices;                        // output tiles along n_in; K slices per tile.  blockIdx.x = tile * slices + slice
    float* part;                                // [tile][slice][32 TA][32 TB]
    float* asum;                                // [tile][slice][2][32 TA]: per lane half, the column sums of dy over the slice (tiles with tb = 0)
};

template <int OFF>
__device__ __forceinline__ void lds_read4_async(float& v, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 4 == 0, "ds_read_b32 offset field");
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

// OCC: workgroups per CU the kernel is built for (2: half the LDS ring each, <= 256 registers -- the small tiles, whose barrier and
// hand-over gaps a second workgroup fills)
template <int SA, int SB, int WA, int WB, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_tn_kernel(const TnArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)       // (host pass: launch stub only -- see gemm_nt.hip)
    static_assert(WA * WB == 4, "four waves");
    constexpr int TA = SA * WA, TB = SB * WB, NL = TA + TB;
    constexpr int BUF = NL * 4096;
    constexpr int D = TN_LDS_BUDGET / OCC / BUF < 4 ? TN_LDS_BUDGET / OCC / BUF : 4;
    static_assert(D >= 2 && (D - 1) * NL <= 63, "ring depth / vmcnt range");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = w / WB, wb = w % WB;
    const int tile = blockIdx.x / a.slices, slice = blockIdx.x % a.slices;
    const int ta = tile / a.tiles_b, tb = tile % a.tiles_b;
    const int64_t per = (a.chunks + a.slices - 1) / a.slices;
    const int64_t c0 = (int64_t)slice * per;
    const int n = (int)(a.chunks - c0 < per ? (a.chunks - c0 > 0 ? a.chunks - c0 : 0) : per);      // chunks of this slice

    // ---- DMA: piece (block j, samples 8 w .. 8 w + 7): lane l fetches the 16 bytes at unit l & 7 of sample 8 w + (l >> 3), landing at
    // the piece's byte 16 l = row (l >> 3), unit l & 7: the tile is row-major [block][32 samples][128 B].
    // The chunks of a slice are requested strictly in order: running pointers, the floats left in each allocation, the samples left.
    const int drow = 8 * w + (lane >> 3);
    const int voA = (int)(drow * a.lda * 4) + ((lane & 7) << 4), voB = (int)(drow * a.ldb * 4) + ((lane & 7) << 4);
    const float* pa = a.A + (int64_t)ta * TA * 32 + c0 * 32 * a.lda;
    const float* pb = a.B + (int64_t)tb * TB * 32 + c0 * 32 * a.ldb;
    int64_t a_left = a.a_floats - ((int64_t)ta * TA * 32 + c0 * 32 * a.lda), b_left = a.b_floats - ((int64_t)tb * TB * 32 + c0 * 32 * a.ldb);
    int64_t rows_left = a.M - c0 * 32;
    auto bound = [](int64_t want, int64_t have) { const int64_t b = want < have ? want : have; return b < 0 ? (int64_t)0 : (b < 0x1fffffff ? b : (int64_t)0x1fffffff); };
    auto issue_chunk = [&](unsigned slot_byte) __attribute__((always_inline)) {               // the slice's next chunk into a ring slot
        const int64_t rows = rows_left < 32 ? rows_left : 32;                                 // samples beyond M read as 0
        const rsrc_t rsA = uniform_rsrc(pa, bound(rows * a.lda, a_left));
        const rsrc_t rsB = uniform_rsrc(pb, bound(rows * a.ldb, b_left));
        float* const dst = lds + (slot_byte + w * 1024) / 4;
#pragma unroll
        for (int j = 0; j < TA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (DMN_LAS void*)(dst + j * 1024), 16, voA, j * 128, 0, 0);
#pragma unroll
        for (int j = 0; j < TB; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (DMN_LAS void*)(dst + (TA + j) * 1024), 16, voB, j * 128, 0, 0);
        pa += 32 * a.lda; pb += 32 * a.ldb;
        a_left -= 32 * a.lda; b_left -= 32 * a.ldb;
        rows_left -= 32;
    };

    // ---- operand reads: block's byte 128 (2 s + half) + 4 li of k-step s
    const unsigned rdA = lds_addr(lds) + (wa * SA) * 4096 + half * 128 + li * 4;
    const unsigned rdB = lds_addr(lds) + (TA + wb * SB) * 4096 + half * 128 + li * 4;
    float ra[2][SA][16], rb[2][SB][16];
    auto read_step = [&](auto pc, auto sc, unsigned slot_byte) __attribute__((always_inline)) {   // k-step s of a chunk into register set p
        constexpr int p = decltype(pc)::value, s = decltype(sc)::value;
        static_for<SA>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_read4_async<i * 4096 + s * 256>(ra[p][i][s], rdA + slot_byte); });
        static_for<SB>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_read4_async<i * 4096 + s * 256>(rb[p][i][s], rdB + slot_byte); });
    };

    f32x16 acc[SA][SB];
    float cs[SA];
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        cs[i] = 0.f;
#pragma unroll
        for (int j = 0; j < SB; ++j) acc[i][j] = (f32x16)(0.f);
    }

    if (n > 0) {
        // prologue: the first D chunks on their way, chunk 0 landed, its operands requested
        for (int c = 0; c < D && c < n; ++c) issue_chunk(c * BUF);
        if (n >= D) __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 1) * NL) & 15) | ((((D - 1) * NL) >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        static_for<16>([&](auto sc) { read_step(std::integral_constant<int, 0>{}, sc, 0u); });

        unsigned sb = 0;                                                                       // slot of the chunk being multiplied
        auto body = [&](auto pc, auto ahead_c, int c) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value;
            constexpr bool read_ahead = decltype(ahead_c)::value;      // (the slice's odd last chunk: nothing follows it)
            const unsigned nb = sb + BUF == (unsigned)(D * BUF) ? 0u : sb + BUF;
            // hand-over: chunk c + 1 has landed (the D - 2 groups behind it may still fly -- when that many were issued), this wave's
            // reads of chunk c have returned; behind the barrier both hold for every wave: slot sb is free, slot nb is readable
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0) (unconditional, its own instruction: scripts/check_asm_hazard.py walks every path)
            if (c + D - 1 < n) __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 2) * NL) & 15) | ((((D - 2) * NL) >> 4) << 14));
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < SA; ++i)
#pragma unroll
                for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(ra[p][i][s]));
#pragma unroll
            for (int i = 0; i < SB; ++i)
#pragma unroll
                for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(rb[p][i][s]));
            if (c + D < n) issue_chunk(sb);
            static_for<16>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
#pragma unroll
                for (int i = 0; i < SA; ++i)
#pragma unroll
                    for (int j = 0; j < SB; ++j) acc[i][j] = mfma32(ra[p][i][s], rb[p][j][s], acc[i][j]);
#pragma unroll
                for (int i = 0; i < SA; ++i) cs[i] += ra[p][i][s];
                if constexpr (read_ahead) read_step(std::integral_constant<int, 1 - p>{}, sc, nb);   // (behind an even last chunk: a stale slot, never used)
                __builtin_amdgcn_sched_barrier(0);
            });
            sb = nb;
        };
        int c = 0;
#pragma nounroll
        for (; c + 1 < n; c += 2) {
            body(std::integral_constant<int, 0>{}, std::true_type{}, c);
            body(std::integral_constant<int, 1>{}, std::true_type{}, c + 1);
        }
        if (c < n) body(std::integral_constant<int, 0>{}, std::false_type{}, c);
        // the last body's read-ahead (a stale slot, never used) is still in flight: its registers are dead to the compiler, which hands
        // them to the epilogue below -- retire the reads first
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // ---- this slice's partial tile (a lane holds column li of its block, rows (r & 3) + 8 (r >> 2) + 4 half) and column sums
    float* const P = a.part + ((int64_t)tile * a.slices + slice) * (TA * 32) * (TB * 32);
#pragma unroll
    for (int i = 0; i < SA; ++i)
#pragma unroll
        for (int j = 0; j < SB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wa * SA + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                P[row * (TB * 32) + (wb * SB + j) * 32 + li] = acc[i][j][r];
            }
    if (wb == 0 && tb == 0) {
        float* const S = a.asum + ((int64_t)ta * a.slices + slice) * 2 * (TA * 32);
#pragma unroll
        for (int i = 0; i < SA; ++i) S[half * (TA * 32) + (wa * SA + i) * 32 + li] = cs[i];
    }
#else
    (void)a;
#endif
}

struct TnReduceArgs {
    const float* part; const float* asum;
    int slices, tiles_b, ta32, tb32;            // ta32 / tb32: rows / columns of an output tile
    int n_out, n_in;
    float* dW; int64_t ldw;
    float* db;                                  // nullable
};

// dW[i][j] = sum over the slices of the tile partials, db[i] likewise over slices and the two lane halves -- in a fixed order: a
// workgroup owns 32 consecutive outputs; thread (j, g) adds the terms g, g + 8, g + 16, ... of output j (8 loads in flight per
// thread: a 128-byte row per slice and wave quarter), then the eight partial sums meet in LDS and are added g = 0 .. 7.
// (One thread per output walked all 256 slices alone: 57 us for a 128 x 128 gradient -- a third of its product's time.)
__global__ __launch_bounds__(256) void reduce_tn_kernel(const TnReduceArgs a) {
    __shared__ float sh[8][33];
    const int j = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t e = (int64_t)blockIdx.x * 32 + j;
    const int64_t total = (int64_t)a.n_out * a.n_in;
    const float* p = nullptr;
    int64_t stride = 0;
    int terms = 0;
    if (e < total) {
        const int i = (int)(e / a.n_in), jj = (int)(e % a.n_in);
        const int tile = (i / a.ta32) * a.tiles_b + jj / a.tb32;
        stride = (int64_t)a.ta32 * a.tb32;
        p = a.part + (int64_t)tile * a.slices * stride + (int64_t)(i % a.ta32) * a.tb32 + jj % a.tb32;
        terms = a.slices;
    } else if (a.db && e < total + a.n_out) {
        const int i = (int)(e - total);
        stride = a.ta32;                                                   // term 2 slice + half
        p = a.asum + (int64_t)(i / a.ta32) * a.slices * 2 * a.ta32 + i % a.ta32;
        terms = 2 * a.slices;
    }
    float s = 0.f;
    int k = g;
    for (; k + 56 < terms; k += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + 8 * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < terms; k += 8) s += p[(int64_t)k * stride];
    sh[g][j] = s;
    __syncthreads();
    if (g == 0 && terms > 0) {
        float t = sh[0][j];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += sh[u][j];
        if (e < total) a.dW[(e / a.n_in) * a.ldw + e % a.n_in] = t;
        else a.db[e - total] = t;
    }
}

struct TnShape { int sa, sb, wa, wb; };
// blocks per wave along one dimension of the 2 x 2 wave grid for nblk 32-column blocks: the tile (2, 4 or 6 blocks) that needs the fewest
// tiles (every extra tile re-reads the OTHER operand), among those the one that pads least
int tn_per_wave(int nblk) {
    int best = 1, best_tiles = 1 << 30, best_pad = 1 << 30;
    for (int s = 3; s >= 1; --s) {
        const int t = 2 * s, tiles = (nblk + t - 1) / t, pad = tiles * t;
        if (tiles < best_tiles || (tiles == best_tiles && pad < best_pad)) { best = s; best_tiles = tiles; best_pad = pad; }
    }
    return best;
}
TnShape tn_shape(int n_out, int n_in) {
    const int na = (n_out + 31) / 32, nb = (n_in + 31) / 32;
    if (na == 1) return {1, nb <= 4 ? 1 : 2, 1, 4};
    if (nb == 1) return {na <= 4 ? 1 : 2, 1, 4, 1};
    return {tn_per_wave(na), tn_per_wave(nb), 2, 2};
}

// two workgroups per CU where both fit: <= 4 accumulator blocks per wave and a chunk of <= 10 blocks (a 2-deep ring in 80 KiB)
int tn_occupancy(const TnShape& s) {
#if defined(DMN_TN_OCC1)
    return 1;
#else
    return s.sa * s.sb <= 4 && s.sa * s.wa + s.sb * s.wb <= 10 ? 2 : 1;
#endif
}

struct TnPlan { TnShape sh; int occ, ta, tb, tiles_a, tiles_b, slices; int64_t chunks, part_floats, asum_floats; };
int tn_plan(int n_out, int n_in, int64_t M, TnPlan* p) {
    int dev = 0, cus = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return dmn_fail_hip(e, "gemm_tn: hipGetDevice");
    if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess || cus < 1)
        return dmn_fail_hip(e, "gemm_tn: hipDeviceGetAttribute");
    p->sh = tn_shape(n_out, n_in);
    p->ta = p->sh.sa * p->sh.wa;
    p->tb = p->sh.sb * p->sh.wb;
    p->tiles_a = ((n_out + 31) / 32 + p->ta - 1) / p->ta;
    p->tiles_b = ((n_in + 31) / 32 + p->tb - 1) / p->tb;
    p->chunks = (M + 31) / 32;
    p->occ = tn_occupancy(p->sh);
    int64_t s = (int64_t)cus * p->occ / (p->tiles_a * p->tiles_b);
    if (s < 1) s = 1;
    if (s > p->chunks) s = p->chunks;
    p->slices = (int)s;
    p->part_floats = (int64_t)p->tiles_a * p->tiles_b * p->slices * (p->ta * 32) * (p->tb * 32);
    p->asum_floats = (int64_t)p->tiles_a * p->slices * 2 * (p->ta * 32);
    return DMNERF_OK;
}

template <int SA, int SB, int WA, int WB, int OCC>
int launch_tn_occ(const TnArgs& a, int blocks, hipStream_t stream) {
    constexpr int NL = SA * WA + SB * WB;
    constexpr int D = TN_LDS_BUDGET / OCC / (NL * 4096) < 4 ? TN_LDS_BUDGET / OCC / (NL * 4096) : 4;
    constexpr int lds_bytes = D * NL * 4096;
    static DmnOncePerDevice once;
    if (hipError_t e = once.run([] { return hipFuncSetAttribute((const void*)gemm_tn_kernel<SA, SB, WA, WB, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); });
        e != hipSuccess)
        return dmn_fail_hip(e, "gemm_tn: hipFuncSetAttribute");
    hipLaunchKernelGGL((gemm_tn_kernel<SA, SB, WA, WB, OCC>), dim3((unsigned)blocks), dim3(256), lds_bytes, stream, a);
    return dmn_check_launch("gemm_tn");
}
template <int SA, int SB, int WA, int WB>
int launch_tn(const TnArgs& a, int blocks, int occ, hipStream_t stream) {
    if constexpr (SA * SB <= 4 && SA * WA + SB * WB <= 10) {
        if (occ == 2) return launch_tn_occ<SA, SB, WA, WB, 2>(a, blocks, stream);
    }
    return launch_tn_occ<SA, SB, WA, WB, 1>(a, blocks, stream);
}

}  // namespace

extern "C" int64_t dmnerf_gemm_tn_ws_floats(int n_out, int n_in, int64_t M) {
    if (n_out < 1 || n_in < 1 || M < 1) return 0;
    TnPlan p;
    if (tn_plan(n_out, n_in, M, &p) != DMNERF_OK) return -1;
    return p.part_floats + p.asum_floats;
}

extern "C" int dmnerf_gemm_tn(const float* d_dy, int64_t ldy, int64_t dy_floats, int n_out, const float* d_x, int64_t ldx, int64_t x_floats,
                              int n_in, int64_t M, float* d_dW, int64_t ldw, float* d_db, float* d_ws, int64_t ws_floats, void* stream) {
    if (M < 0 || n_out < 1 || n_in < 1 || ldw < n_in) return dmn_fail(DMNERF_E_ARG, "gemm_tn: bad sizes M=%lld n_out=%d n_in=%d ldw=%lld", (long long)M, n_out, n_in, (long long)ldw);
    if (!d_dy || !d_x || !d_dW || !d_ws) return dmn_fail(DMNERF_E_ARG, "gemm_tn: null pointer");
    if (ldy % 4 || ldx % 4 || ((uintptr_t)d_dy & 15) || ((uintptr_t)d_x & 15))
        return dmn_fail(DMNERF_E_ARG, "gemm_tn: operand rows must be 16-byte aligned (ldy=%lld ldx=%lld)", (long long)ldy, (long long)ldx);
    if (ldy * 4 * 32 > 0x3fffffffLL || ldx * 4 * 32 > 0x3fffffffLL) return dmn_fail(DMNERF_E_ARG, "gemm_tn: row stride too large");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {                                           // no samples: zero gradients
        for (int i = 0; i < n_out; ++i)
            if (hipError_t e = hipMemsetAsync(d_dW + (int64_t)i * ldw, 0, (size_t)n_in * 4, st); e != hipSuccess) return dmn_fail_hip(e, "gemm_tn: hipMemsetAsync");
        if (d_db)
            if (hipError_t e = hipMemsetAsync(d_db, 0, (size_t)n_out * 4, st); e != hipSuccess) return dmn_fail_hip(e, "gemm_tn: hipMemsetAsync");
        return DMNERF_OK;
    }
    TnPlan p;
    if (int rc = tn_plan(n_out, n_in, M, &p); rc != DMNERF_OK) return rc;
    if (ws_floats < p.part_floats + p.asum_floats)
        return dmn_fail(DMNERF_E_ARG, "gemm_tn: workspace of %lld floats, %lld needed", (long long)ws_floats, (long long)(p.part_floats + p.asum_floats));
    TnArgs a{};
    a.A = d_dy; a.lda = ldy; a.a_floats = dy_floats;
    a.B = d_x; a.ldb = ldx; a.b_floats = x_floats;
    a.M = M; a.chunks = p.chunks; a.tiles_b = p.tiles_b; a.slices = p.slices;
    a.part = d_ws; a.asum = d_ws + p.part_floats;
    const int blocks = p.tiles_a * p.tiles_b * p.slices;
    const TnShape& s = p.sh;
    int rc = DMNERF_OK;
    if (s.wa == 1) rc = s.sb == 1 ? launch_tn<1, 1, 1, 4>(a, blocks, p.occ, st) : launch_tn<1, 2, 1, 4>(a, blocks, p.occ, st);
    else if (s.wb == 1) rc = s.sa == 1 ? launch_tn<1, 1, 4, 1>(a, blocks, p.occ, st) : launch_tn<2, 1, 4, 1>(a, blocks, p.occ, st);
    else {
#define DMN_TN_CASE(SA_, SB_) case SA_ * 4 + SB_: rc = launch_tn<SA_, SB_, 2, 2>(a, blocks, p.occ, st); break;
        switch (s.sa * 4 + s.sb) {
            DMN_TN_CASE(1, 1) DMN_TN_CASE(1, 2) DMN_TN_CASE(1, 3)
            DMN_TN_CASE(2, 1) DMN_TN_CASE(2, 2) DMN_TN_CASE(2, 3)
            DMN_TN_CASE(3, 1) DMN_TN_CASE(3, 2) DMN_TN_CASE(3, 3)
            default: return dmn_fail(DMNERF_E_ARG, "gemm_tn: no kernel for %d x %d blocks per wave", s.sa, s.sb);
        }
#undef DMN_TN_CASE
    }
    if (rc != DMNERF_OK) return rc;
    TnReduceArgs r{};
    r.part = a.part; r.asum = a.asum; r.slices = p.slices; r.tiles_b = p.tiles_b; r.ta32 = p.ta * 32; r.tb32 = p.tb * 32;
    r.n_out = n_out; r.n_in = n_in; r.dW = d_dW; r.ldw = ldw; r.db = d_db;
    const int64_t total = (int64_t)n_out * n_in + (d_db ? n_out : 0);
    hipLaunchKernelGGL(reduce_tn_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, r);
    return dmn_check_launch("gemm_tn reduce");
}
