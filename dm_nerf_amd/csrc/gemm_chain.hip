// gemm_chain.hip -- the TRUNK of a narrow DM_NeRF (network shapes other than the shipped one, W = 32 .. 160) as ONE launch in inference:
// the activations of a 128-sample tile stay in LDS from layer to layer, only the weights stream.
//
// Layer by layer (gemm_nt.hip) a W = 128 layer costs one HBM round trip of its activations -- 0.8 GB per 4096 x 192 samples against
// 0.16 ms of MFMA work -- and a 128 -> 128 layer ran at 0.57 of the f32 roof however its loop was scheduled
// (profiles/r06/gemm_nt_ablation_r06.txt).  Here:
//   * ACT [chunk b][wave w][32 rows][128 B]: the current layer's input, wave-private (wave w owns samples 32 w .. 32 w + 31 of the tile
//     for the whole trunk), in exactly the row-swizzled block form gemm_nt's DMA produces -- so the K loop reads its activation operand
//     with the same ds_read_b128, only from ACT instead of a ring slot.  The WEIGHTS are the MFMA's row operand here: a lane's
//     accumulator registers are then 4 consecutive features of ONE sample (its li), and after the loop the wave writes relu(acc) back
//     into ACT as 16-byte units (4 ds_write_b128 per block and lane, one v_max_i32 per value) -- no barrier, no HBM;
//   * AUX: the tile's positional encoding (dm_nerf.py:85-87: layer 0 reads it, the layer after a skip reads [h, pts]), fetched once
//     per tile by LDS-DMA, each wave its own 32 rows; the NEXT tile's rows are requested as soon as the last layer that reads them is done;
//   * the weights of all layers are ONE stream of 32-k chunks (NBB blocks of 32 rows x 128 B each) through a D-deep ring shared by the
//     four waves, continuous across layers and tiles (gemm_nt's protocol: hand-over at the start of a chunk's last round, refills
//     behind it); every layer's bias sits in an 8-KiB LDS table staged once per workgroup;
//   * the last layer's result leaves through ACT as 1-KiB stores (8 rows x 128 B), like gemm_nt's staged epilogue.
// v_mfma_f32_32x32x2_f32 throughout: exact f32, bias as the accumulator's start value -- the same arithmetic as the layer-by-layer path
// up to the summation order inside a chunk (none: the chunk order and the k order inside a chunk are gemm_nt's).
// Roofline: MFMA f32 (the only HBM traffic is the encoding in and the trunk's output out).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "mlp_common.h"

using namespace dmn;

namespace {

constexpr int CH_MAX_LAYERS = DMNERF_CHAIN_MAX_LAYERS;
constexpr int CH_MAX_NBB = 5;             // W <= 160: ACT (16 KiB per out-block) + AUX + the weight ring must fit the CU's 160 KiB
constexpr int CH_LDS_BUDGET = 163840;
constexpr int CH_BIAS_BYTES = 8192;        // every layer's bias, staged once per workgroup (a VMEM load per layer would be waited for with
                                          // vmcnt(0) by the compiler -- i.e. for the youngest weight chunk in flight)

struct ChainLayerDev {
    const float* B;                       // packed weights [32 NBB][ldb] (dmnerf_pack_nt: range 0 = the h columns, range 1 = the pts columns)
    const float* bias;                    // [32 NBB]
    int ldb, kA, kX, relu;                // chunks read from ACT / from AUX
};

struct ChainArgs {
    const float* X; int64_t ldx, x_floats;    // row-padded encoding [M][ldx] (dmnerf_ray_embed), floats to the end of its allocation
    int nx;                               // chunks of the encoding (AUX holds nx x 16 KiB)
    float* out; int64_t ldo;              // the trunk's output h [M][ldo], columns [0, 32 NBB)
    int64_t M;
    int n_layers, last_aux_layer;         // last_aux_layer: the last layer with kX > 0 (the next tile's AUX is requested behind it)
    int total_chunks;                     // sum of kA + kX over the layers
    ChainLayerDev L[CH_MAX_LAYERS];
    long long* trace;                     // diagnostic builds only (scripts/diag_chain.py): shader-clock stamps of wave 0, tile `CH_TRACE_TILE`
};
#if defined(DMN_CH_TRACE)
#define CH_TRACE_TILE 3
#define DMN_CH_STAMP(k) do { if (args()->trace && threadIdx.x == 0 && tile_seq == CH_TRACE_TILE) args()->trace[(int64_t)blockIdx.x * 64 + (k)] = (long long)clock64(); } while (0)
#else
#define DMN_CH_STAMP(k) do { } while (0)
#endif

// D: slots of the weight ring (2 .. 4; the host derives it from the width and the encoding's size -- a template parameter because the
// hand-over's s_waitcnt takes its count as an immediate, and as a run-time value every chunk walked a ladder of branches for it)
template <int NBB, int D>
__global__ __launch_bounds__(256) void chain_kernel(const ChainArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)       // (host pass: launch stub only -- see gemm_nt.hip)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NL = NBB;                         // DMA pieces per wave per weight chunk
    constexpr int BUF = NBB * 4096;                 // bytes per weight chunk
    constexpr int NR = 1 + NBB, NGAP = 4 * NBB;
    typedef const ChainArgs __attribute__((address_space(4))) KArgs;
    auto args = [&]() -> KArgs* { KArgs* p = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(p)); return p; };
    auto fresh_s = [](int x) -> int { asm volatile("" : "+s"(x)); return x; };
    auto fresh_v = [](int x) -> int { asm volatile("" : "+v"(x)); return x; };
    const int tid = threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nx = a.nx;
    // LDS: [bias table: 8 KiB][ACT: NBB x 16 KiB][AUX: nx x 16 KiB][ring: D x BUF]
    float* const btab = lds;
    float* const act = lds + CH_BIAS_BYTES / 4;
    float* const aux = act + NBB * 4096;
    float* const ring = aux + nx * 4096;
    const unsigned ring0 = lds_addr(ring), act0 = lds_addr(act), aux0 = lds_addr(aux);
    const int ntiles = (int)((a.M + 127) / 128);   // (32-bit: see gemm_nt.hip -- a 64-bit `<` is a vector compare and drags the fetch state into VGPRs)
    int tile = blockIdx.x;

    // ---- weight-ring DMA geometry (gemm_nt): wave w owns piece w of every 32-row block; lane l lands at row 8 w + (l >> 3), unit l & 7
    const int drow = 8 * w + (lane >> 3);
    const int dunit = ((lane & 7) ^ ((drow >> 1) & 7)) << 4;
    // ---- AUX DMA geometry: wave w fetches ITS OWN 32 rows of a chunk: piece j = rows 8 j .. 8 j + 7 of block w
    const int xrow = lane >> 3;                                                    // + 8 j
    const int xunit_even = ((lane & 7) ^ ((xrow >> 1) & 7)) << 4;                  // j even: (row >> 1) & 7 = xrow >> 1
    const int xunit_odd = ((lane & 7) ^ (((xrow >> 1) + 4) & 7)) << 4;             // j odd:  + 4
    const int voX_even = (int)(xrow * a.ldx * 4) + xunit_even, voX_odd = (int)(xrow * a.ldx * 4) + xunit_odd;
    auto bound = [](int64_t want, int64_t have) { const int64_t b = want < have ? want : have; return b < 0x1fffffff ? b : (int64_t)0x1fffffff; };
    auto issue_aux = [&](int t) __attribute__((always_inline)) {               // the encoding rows of tile t (this wave's 32) into AUX
        KArgs* q = args();
        const int64_t r0 = (int64_t)t * 128 + 32 * w;
        const int64_t rows = q->M - r0 < 32 ? (q->M - r0 > 0 ? q->M - r0 : 0) : 32;          // rows beyond M read as 0
        const rsrc_t rsX = uniform_rsrc(q->X + (r0 < q->M ? r0 : 0) * q->ldx, rows > 0 ? bound(rows * q->ldx, q->x_floats - r0 * q->ldx) : 0);
        const int ld8 = (int)(8 * q->ldx * 4);
        for (int x = 0; x < nx; ++x) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* dst = aux + x * 4096 + fresh_s(w) * 1024 + j * 256;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (DMN_LAS void*)dst, 16, (j & 1) ? voX_odd : voX_even, j * ld8 + x * 128, 0, 0);
            }
        }
    };

    // ---- the weight stream: (layer fl, chunk fc) is the next request
    int fl = 0, fc = 0, ahead = 0;
    int ftile = tile;
    bool fvalid = true;
    // The FETCH layer's operands live in registers and change once per layer: reading them from the kernarg segment per chunk put four
    // to six scalar-load round trips (each an s_waitcnt the one wave per SIMD sits out) into every 4096-cycle chunk.
    rsrc_t rsB;
    int voB = 0, blkB = 0, fn = 0;                       // descriptor, lane offset, bytes per 32-row block, chunks of the fetch layer
    auto load_fetch_layer = [&]() __attribute__((always_inline)) {
        KArgs* q = args();
        const int ldb = q->L[fl].ldb;
        rsB = uniform_rsrc(q->L[fl].B, (int64_t)NBB * 32 * ldb);
        voB = drow * ldb * 4 + dunit;
        blkB = 32 * ldb * 4;
        fn = q->L[fl].kA + q->L[fl].kX;
    };
    auto issue_weight_piece = [&](unsigned slot_byte, int i) __attribute__((always_inline)) {  // piece i of NL of (fl, fc)
#if defined(DMN_CH_NO_W)      /* diagnostic builds only (scripts/diag_chain.sh): timing without the weight requests; results are wrong */
        return;
#endif
        float* dst = ring + (slot_byte + i * 4096 + fresh_s(w) * 1024) / 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (DMN_LAS void*)dst, 16, voB, i * fresh_s(blkB) + fc * 128, 0, 0);
    };
    auto issue_weights = [&](unsigned slot_byte) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NL; ++i) issue_weight_piece(slot_byte, i);
    };
    auto advance_fetch = [&]() __attribute__((always_inline)) {
        if (++fc == fn) {
            fc = 0;
            if (++fl == args()->n_layers) {
                fl = 0;
                ftile += (int)gridDim.x;
                fvalid = ftile < ntiles;
            }
            load_fetch_layer();
        }
    };
    load_fetch_layer();

    // ---- read geometry: lane (li, half) reads row li of its block, unit (2 t + half) ^ ((li >> 1) & 7) in round t
    unsigned offA[4], offB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned o = li * 128 + ((((2 * t + half) ^ ((li >> 1) & 7))) << 4);
        offA[t] = o + w * 4096;                     // + region base + chunk * 16384
        offB[t] = ring0 + o;                        // + slot
    }
    f32x4 av[2][1], bv[2][NBB];
    auto read_ops_one = [&](auto gc, int buf, unsigned addrA, unsigned addrB) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g == 0) lds_read16_async<0>(av[buf][0], addrA);
        else lds_read16_async<(g - 1) * 4096>(bv[buf][g - 1], addrB);
    };
    // LDS byte address (without the lane part) of the A operand of chunk c of layer l
    auto a_base = [&](int kA, int c) __attribute__((always_inline)) -> unsigned { return c < kA ? act0 + c * 16384 : aux0 + (c - kA) * 16384; };

    // ---- prologue: this tile's encoding, then the first D chunks of the weight stream
    issue_aux(tile);
    for (int sl = 0; sl < D; ++sl)
        if (fvalid) {
            issue_weights(sl * BUF);
            advance_fetch();
            ++ahead;
        }
    --ahead;
    {                                                    // every layer's bias into the LDS table (n_layers x 32 NBB floats)
        const int nl = args()->n_layers;
        for (int l = 0; l < nl; ++l)
            if (tid < NBB * 32) btab[l * NBB * 32 + tid] = args()->L[l].bias[tid];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    static_for<NR>([&](auto gc) { read_ops_one(gc, 0, a_base(args()->L[0].kA, 0) + offA[0], offB[0]); });

    f32x16 acc[NBB];
    unsigned sb = 0;
#if defined(DMN_CH_TRACE)
    int tile_seq = 0;
#endif
#pragma nounroll
    for (;;) {                                           // tiles
        const int n_layers = args()->n_layers;
        DMN_CH_STAMP(0);
#pragma nounroll
        for (int l = 0; l < n_layers; ++l) {
            const int kA = args()->L[l].kA;
            const int nchunk = kA + args()->L[l].kX;
            const int ln = l + 1 < n_layers ? l + 1 : 0;
            const int kA_next = args()->L[ln].kA;
            {
                // accumulator register 4 q + e of block b: feature 32 b + 8 q + 4 half + e (of sample li)
                const float* bt = btab + l * NBB * 32 + fresh_v(4 * half);
#pragma unroll
                for (int b = 0; b < NBB; ++b) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(bt + 32 * b + 8 * q4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[b][4 * q4 + e] = v[e];
                    }
                }
            }
#pragma nounroll
            for (int c = 0; c < nchunk; ++c) {
                const unsigned nb = sb + BUF == (unsigned)(D * BUF) ? 0u : sb + BUF;
                // the stream's next chunk: chunk c + 1 of this layer, or chunk 0 of the next layer (of the next tile's first)
                const bool last_c = c + 1 == nchunk;
                const unsigned an_next = last_c ? a_base(kA_next, 0) : a_base(kA, c + 1);
                const unsigned an_cur = a_base(kA, c);
                unsigned cA[4], cB[4];
#pragma unroll
                for (int t = 1; t < 4; ++t) { cA[t] = an_cur + offA[t]; cB[t] = offB[t] + sb; }
                cA[0] = an_next + offA[0]; cB[0] = offB[0] + nb;
                static_for<4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
#if defined(DMN_CH_TRACE)
                    if (l == 2 && c == 1) DMN_CH_STAMP(48 + r);
#endif
                    lds_wait<0>(av[r & 1]);
#pragma unroll
                    for (int k = 0; k < NBB; ++k) asm volatile("" : "+" DMN_TILE_RC(bv[r & 1][k]));
                    if constexpr (r == 3) {
                        if (ahead == D - 1) {                      // the stream's next chunk has landed: all but the (D - 2) NL youngest requests
                            constexpr int keep = (D - 2) * NL;
                            __builtin_amdgcn_s_waitcnt(0x0F70 | (keep & 15) | ((keep >> 4) << 14));
                        } else {
                            __builtin_amdgcn_s_waitcnt(0x0F70);
                        }
#if !defined(DMN_CH_NO_BAR)
                        __builtin_amdgcn_s_barrier();
#endif
                        asm volatile("" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<NGAP>([&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        constexpr int u = g / NBB, ib = g % NBB;
                        if constexpr (g < NR) {
                            if constexpr (r == 3) {
                                // round 0 of the NEXT chunk: its B operands now (the ring slot has landed); its A operand now too unless it
                                // lives in the ACT this layer's epilogue is about to rewrite (then after that epilogue)
                                if constexpr (g == 0) { if (!last_c) read_ops_one(gc, 0, cA[0], cB[0]); }
                                else read_ops_one(gc, 0, cA[0], cB[0]);
                            } else {
                                read_ops_one(gc, (r + 1) & 1, cA[(r + 1) & 3], cB[(r + 1) & 3]);
                            }
                        }
                        if constexpr (r == 3 && g >= NR && g < NR + NL) {    // (NR + NL = 1 + 2 NBB <= NGAP - 1)
                            if (fvalid) issue_weight_piece(sb, g - NR);   // refill the released slot with the stream's next chunk, a request per gap
                        }
                        if constexpr (r == 3 && g == NR + NL) {           // ... and the stream's bookkeeping under an MFMA as well
                            if (fvalid) advance_fetch();
                            else --ahead;
                        }
                        acc[ib] = mfma32(bv[r & 1][ib][u], av[r & 1][0][u], acc[ib]);   // (weights as the ROW operand: see the epilogue)
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
#if defined(DMN_CH_TRACE)
                if (l == 2 && c == 1) DMN_CH_STAMP(52);
#endif
                sb = nb;
#if defined(DMN_CH_TRACE)
                if (l == 2 && c == 1) DMN_CH_STAMP(53);
#endif
            }
            // ---- layer epilogue: relu(acc) -> ACT (this wave's rows).  Every ds_read of this layer's A operand has returned (the last
            // round's operands were consumed by its MFMAs), so the rows may be rewritten.
            lds_wait<0>(bv[0]);                                    // (the B read-ahead of the next chunk: registers handed back)
            DMN_CH_STAMP(1 + 2 * l);
            {
                KArgs* q = args();
                const int relu = q->L[l].relu;
                const int lo = fresh_s(relu ? 0 : (int)0x80000000);  // (opaque: one v_max_i32 per value instead of a max and a select) relu1 (mlp_common.h) is an integer max with 0; without relu: with INT_MIN
                const int lane_e = fresh_v(lane);
                const int half_e = lane_e >> 5, li_e = lane_e & 31;
                float* const aw = act + w * 1024 + li_e * 32;      // this sample's row (+ block * 4096 floats)
                const int sw = (li_e >> 1) & 7;
#if defined(DMN_CH_NO_EPI)
                if (relu == 77)
#endif
#pragma unroll
                for (int b = 0; b < NBB; ++b) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {               // features 32 b + 8 q4 + 4 half .. + 3: unit 2 q4 + half of the row
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int xi = (int)f2u(acc[b][4 * q4 + e]);
                            v[e] = __uint_as_float((unsigned)(xi > lo ? xi : lo));
                        }
                        *reinterpret_cast<f32x4*>(aw + b * 4096 + (((2 * q4 + half_e) ^ sw) << 2)) = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the encoding is free once the last layer that reads it has run: request the NEXT tile's rows now
                #if !defined(DMN_CH_NO_AUX)
                if (l == q->last_aux_layer && tile + (int)gridDim.x < ntiles) issue_aux(tile + (int)gridDim.x);
#endif
            }
            // the A operand of the stream's next chunk (round 0) now that ACT holds the new activations (at the tile boundary: below)
            if (l + 1 < n_layers) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's ds_writes have landed
                lds_read16_async<0>(av[0][0], a_base(kA_next, 0) + offA[0]);
            }
            DMN_CH_STAMP(2 + 2 * l);
        }
        // ---- the trunk's output: ACT rows -> HBM as 1-KiB stores (lane l: row 8 j + (l >> 3), the 16 bytes at position l & 7 = unit
        // (l & 7) ^ ((row >> 1) & 7) of the row)
        {
            KArgs* q = args();
            const int64_t r0 = (int64_t)tile * 128 + 32 * w;
            const int64_t rows = q->M - r0 < 32 ? q->M - r0 : 32;
#if defined(DMN_CH_NO_OUT)
            if (rows > 77)
#else
            if (rows > 0)
#endif
            {
                const rsrc_t rsC = uniform_rsrc(q->out + r0 * q->ldo, (rows - 1) * q->ldo + NBB * 32);
                const int lane_e = fresh_v(lane);
                const int rowl = lane_e >> 3, pos = lane_e & 7;
                const float* const aw = act + w * 1024;
                const int rowB = fresh_v((int)(q->ldo * 4));
#pragma unroll
                for (int b = 0; b < NBB; ++b) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = 8 * j + rowl;
                        const int unit = pos ^ ((row >> 1) & 7);
                        const f32x4 v = *reinterpret_cast<const f32x4*>(aw + b * 4096 + row * 32 + pos * 4);
                        u32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = f2u(v[e]);
                        __builtin_amdgcn_raw_buffer_store_b128(o, rsC, row * rowB + (32 * b + 4 * unit) * 4, 0, 0);
                    }
                }
            }
        }
        DMN_CH_STAMP(40);
        const int next = tile + (int)gridDim.x;
        if (next >= ntiles) break;
        tile = next;
        // the next tile's encoding has landed (requested behind the last layer that read this tile's; when that IS the last layer its
        // latency is exposed here, once per tile) -- and with it everything older; then layer 0's first A operand
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        lds_read16_async<0>(av[0][0], a_base(args()->L[0].kA, 0) + offA[0]);
        DMN_CH_STAMP(41);
#if defined(DMN_CH_TRACE)
        ++tile_seq;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
    (void)a;
#endif
}

template <int NBB, int D>
int launch_chain(const ChainArgs& a, hipStream_t stream) {
    const int lds_bytes = CH_LDS_BUDGET;
    static DmnOncePerDevice once;
    if (hipError_t e = once.run([] { return hipFuncSetAttribute((const void*)chain_kernel<NBB, D>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BUDGET); });
        e != hipSuccess)
        return dmn_fail_hip(e, "mlp_chain: hipFuncSetAttribute");
    int dev = 0, cus = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return dmn_fail_hip(e, "mlp_chain: hipGetDevice");
    if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess || cus < 1)
        return dmn_fail_hip(e, "mlp_chain: hipDeviceGetAttribute");
    const int64_t ti = (a.M + 127) / 128;
    hipLaunchKernelGGL((chain_kernel<NBB, D>), dim3((unsigned)(ti < cus ? ti : cus)), dim3(256), lds_bytes, stream, a);
    return dmn_check_launch("mlp_chain");
}

}  // namespace

#if defined(DMN_CH_TRACE)
static long long* g_dmn_ch_trace = nullptr;
extern "C" int dmnerf_mlp_chain_set_trace(int64_t* d_ticks) { g_dmn_ch_trace = (long long*)d_ticks; return 0; }
#endif

extern "C" int dmnerf_mlp_chain_supported(int width, int x_cols) {
    if (width < 32 || width % 32 || width / 32 > CH_MAX_NBB || x_cols < 1) return 0;
    const int nbb = width / 32, nx = (x_cols + 31) / 32;
    return (CH_LDS_BUDGET - CH_BIAS_BYTES - (nbb + nx) * 16384) / (nbb * 4096) >= 2 ? 1 : 0;
}

extern "C" int dmnerf_mlp_chain(const float* d_x, int64_t ldx, int64_t x_floats, int x_cols, const dmnerf_chain_layer* layers, int n_layers,
                                int width, float* d_out, int64_t ldo, int64_t M, void* stream) {
    if (M < 0 || n_layers < 1 || n_layers > CH_MAX_LAYERS || n_layers * (width / 32) * 128 > CH_BIAS_BYTES) return dmn_fail(DMNERF_E_ARG, "mlp_chain: bad sizes M=%lld layers=%d", (long long)M, n_layers);
    if (!dmnerf_mlp_chain_supported(width, x_cols)) return dmn_fail(DMNERF_E_ARG, "mlp_chain: width %d / encoding %d columns not supported", width, x_cols);
    if (M == 0) return DMNERF_OK;
    if (!d_x || !layers || !d_out) return dmn_fail(DMNERF_E_ARG, "mlp_chain: null pointer");
    if (ldx % 4 || ((uintptr_t)d_x & 15) || ldo % 4 || ((uintptr_t)d_out & 15) || ldo < width)
        return dmn_fail(DMNERF_E_ARG, "mlp_chain: rows must be 16-byte aligned (ldx=%lld ldo=%lld)", (long long)ldx, (long long)ldo);
    if (ldx * 4 * 128 > 0x3fffffffLL || ldo * 4 * 128 > 0x3fffffffLL) return dmn_fail(DMNERF_E_ARG, "mlp_chain: row stride too large");
    const int nbb = width / 32, nx = (x_cols + 31) / 32;
    ChainArgs a{};
    a.X = d_x; a.ldx = ldx; a.x_floats = x_floats; a.nx = nx; a.out = d_out; a.ldo = ldo; a.M = M; a.n_layers = n_layers;
    a.last_aux_layer = -1; a.total_chunks = 0;
#if defined(DMN_CH_TRACE)
    a.trace = g_dmn_ch_trace;
#endif
    for (int l = 0; l < n_layers; ++l) {
        const dmnerf_chain_layer& s = layers[l];
        const int kA = s.from_act ? nbb : 0, kX = s.from_x ? nx : 0;
        if (!s.d_B || !s.d_bias || kA + kX == 0 || s.ldb != 32 * (kA + kX) || ((uintptr_t)s.d_B & 15))
            return dmn_fail(DMNERF_E_ARG, "mlp_chain: layer %d: bad operands (ldb=%d, expected %d)", l, s.ldb, 32 * (kA + kX));
        if (l == 0 && kA) return dmn_fail(DMNERF_E_ARG, "mlp_chain: layer 0 has no activations to read");
        a.L[l] = ChainLayerDev{s.d_B, s.d_bias, s.ldb, kA, kX, s.relu};
        if (kX) a.last_aux_layer = l;
        a.total_chunks += kA + kX;
    }
    hipStream_t st = (hipStream_t)stream;
    const int slots = (CH_LDS_BUDGET - CH_BIAS_BYTES - (nbb + nx) * 16384) / (nbb * 4096);        // >= 2 (dmnerf_mlp_chain_supported)
    const int depth = slots < 4 ? slots : 4;
#define DMN_CH_CASE(N)                                                                                           \
    case N: return depth == 2 ? launch_chain<N, 2>(a, st) : depth == 3 ? launch_chain<N, 3>(a, st) : launch_chain<N, 4>(a, st);
    switch (nbb) {
        DMN_CH_CASE(1)
        DMN_CH_CASE(2)
        DMN_CH_CASE(3)
        DMN_CH_CASE(4)
        DMN_CH_CASE(5)
        default: return dmn_fail(DMNERF_E_ARG, "mlp_chain: unsupported width %d", width);
    }
#undef DMN_CH_CASE
}
