// layout.h -- weight "blob" layout shared by the packer (host) and the fused MLP kernels (device).
//
// The fused kernel keeps activations in MFMA accumulator layout and feeds them straight back
// as the B operand of the next layer.  For v_mfma_f32_32x32x2_f32:
//   A[i][k]: lane l supplies row i = l&31,  k = l>>5          (one f32 VGPR)
//   B[k][j]: lane l supplies col j = l&31,  k = l>>5          (one f32 VGPR)
//   C[i][j]: lane l holds   col j = l&31,  rows (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15
// With W (out x in) as A and X^T (in x samples) as B, a lane's 16 accumulator registers of
// out-block b are 16 features of ITS sample; register r of block b pairs feature
// f0 = 32b + (r&3) + 8*(r>>2) (lanes 0-31) with f0+4 (lanes 32-63) -- exactly a k-pair of the
// next layer.  So "k-pair p = 16b + r"  <->  input features cfeat(p, half).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DMN_HD __host__ __device__
#else
#define DMN_HD
#endif

namespace dmn {

// accumulator-layout k order: k-pair p (0..), half (lane>>5) -> feature index
DMN_HD constexpr int cfeat(int p, int half) {
    return 32 * (p >> 4) + ((p & 15) & 3) + 8 * ((p & 15) >> 2) + 4 * half;
}

// positional-encoding k order for L frequencies: pairs 0,1 = (x,y),(z,pad); pair 2+3k+c =
// (sin(2^k x_c), cos(2^k x_c)).  Returns the column in Embedder.embed's output
// (networks/dm_nerf.py:37: [x | sin f0 | cos f0 | sin f1 | ...], blocks of 3) or -1 (zero pad).
DMN_HD constexpr int pefeat(int p, int half, int L) {
    if (p == 0) return half;                 // x, y
    if (p == 1) return half ? -1 : 2;        // z, pad
    int q = p - 2;
    if (q >= 3 * L) return -1;
    int k = q / 3, c = q % 3;
    return 3 + 6 * k + 3 * half + c;
}

constexpr int W = 256;        // trunk width
constexpr int HW = 128;       // head hidden width (W/2)
constexpr int POS_CH = 63, DIR_CH = 27, POS_L = 10, DIR_L = 4;
constexpr int POS_KP = 32;    // k-pairs of the padded position encoding (64 slots)
constexpr int DIR_KP = 16;    // k-pairs of the padded direction encoding (32 slots)
constexpr int NSTAGE = 9;     // 256->256 GEMMs sharing one code body: L1..L4, L5(h part), L6, L7, rgb_feature, ins_feature

// A gemm segment of NKG k-groups (4 k-pairs each) x OB out-blocks (32 rows each):
//   seg[((g*OB + ob)*64 + lane)*4 + kk] = Wmat[ob*32 + (lane&31)][kmap(4g+kk, lane>>5)]
constexpr int64_t seg_floats(int nkg, int ob) { return (int64_t)nkg * ob * 256; }
// bias segment: seg[(ob*2 + half)*16 + r] = bias[32ob + (r&3) + 8(r>>2) + 4half]
constexpr int64_t bias_floats(int ob) { return (int64_t)ob * 32; }

// Forward blob = [table | weight stream | 2 dummy quarters].
//   table  (TAB_FLOATS, staged once per workgroup in LDS): every bias in accumulator order plus the
//          two VALU heads (density_linear, rgb_linear);
//   stream: the GEMM segments in EXACT consumption order, cut into "quarters" of 64 KiB (one LDS
//          ring slot); segments shorter than a quarter are padded to one, so the stream pointer
//          always advances by 64 KiB and the DMA schedule is completely static:
//            w0 | st0..st4 (4 quarters each) | w5pe | st5 st6 st7 | rgbh (2) | rgbh_dir | st8 | insh (2) | inso
//          st = the nine 256->256 stages L1..L4, L5(h part), L6, L7, rgb_feature, ins_feature.
constexpr int QUARTER_FLOATS = 16384;    // 64 KiB: 8 k-groups x 8 out-blocks x 1 KiB
constexpr int TAB_FLOATS = 4096;         // 16 KiB
constexpr int N_QUARTERS = 1 + 5 * 4 + 1 + 3 * 4 + 3 + 4 + 3;   // 44

struct BlobLayout {
    int C, OBI;                   // logits, out-blocks of the ins_linear head
    // table
    int64_t b0, b_stage;          // bias of mlps.0; NSTAGE x bias(8)
    int64_t b_rgbh, b_insh, b_inso;
    int64_t w_den, b_den;         // density_linear on VALU: [half][128] + bias (padded to 4)
    int64_t w_rgbo, b_rgbo;       // rgb_linear on VALU: [c][half][64] + bias[3] (padded to 4)
    // stream (float offsets from the blob start; every one a multiple of QUARTER_FLOATS past `stream`)
    int64_t stream;
    int64_t w0;                   // mlps.0: PE k-order, NKG=8, OB=8 (1 quarter)
    int64_t w_stage_lo;           // stages 0..4 (4 quarters each)
    int64_t w5pe;                 // mlps.5 columns 256..318 (1 quarter)
    int64_t w_stage_mid;          // stages 5..7
    int64_t w_rgbh, w_rgbh_dir;   // rgb_feature_linears.0: (NKG=32,OB=4) 2 quarters + dirs (NKG=4,OB=4) 1 padded quarter
    int64_t w_stage_hi;           // stage 8 (ins_feature)
    int64_t w_insh;               // ins_feature_linears.0: 2 quarters
    int64_t w_inso;               // ins_linear: NKG=16, OB=OBI, 1 padded quarter
    int64_t total;
};

DMN_HD inline int64_t stage_off(const BlobLayout& L, int st) {
    return st < 5 ? L.w_stage_lo + st * seg_floats(32, 8) : (st < 8 ? L.w_stage_mid + (st - 5) * seg_floats(32, 8) : L.w_stage_hi);
}

// fused = the inference-only blob with rgb_feature_linear / ins_feature_linear folded into the hidden layers that
// follow them (no activation in between, dm_nerf.py:89,96): the stream simply has no st7 / st8 quarters.
DMN_HD inline BlobLayout make_layout(int ins_num, bool fused = false) {
    BlobLayout L;
    L.C = ins_num + 1;
    L.OBI = (L.C + 31) / 32;
    int64_t o = 0;
    L.b0 = o; o += bias_floats(8);
    L.b_stage = o; o += NSTAGE * bias_floats(8);
    L.b_rgbh = o; o += bias_floats(4);
    L.b_insh = o; o += bias_floats(4);
    L.b_inso = o; o += bias_floats(L.OBI);
    L.w_den = o; o += 256;
    L.b_den = o; o += 4;
    L.w_rgbo = o; o += 3 * 2 * 64;
    L.b_rgbo = o; o += 4;
    o = TAB_FLOATS;
    L.stream = o;
    L.w0 = o; o += QUARTER_FLOATS;
    L.w_stage_lo = o; o += 5 * seg_floats(32, 8);
    L.w5pe = o; o += QUARTER_FLOATS;
    L.w_stage_mid = o; o += (fused ? 2 : 3) * seg_floats(32, 8);
    L.w_rgbh = o; o += seg_floats(32, 4);
    L.w_rgbh_dir = o; o += QUARTER_FLOATS;
    L.w_stage_hi = o; o += fused ? 0 : seg_floats(32, 8);
    L.w_insh = o; o += seg_floats(32, 4);
    L.w_inso = o; o += QUARTER_FLOATS;
    o += 2 * QUARTER_FLOATS;            // the DMA engine always runs two quarters ahead: zero-filled landing zone
    L.total = o;
    return L;
}

// ---- split-bf16 ("bf16x3") inference blob (opt-in; mlp_split.hip) -----------------------------------------
// v_mfma_f32_32x32x16_bf16: A[i][k]: lane l holds row i = l & 31 and the 8 k-slots 8 (l >> 5) + q, q = 0..7;
// B likewise with col j = l & 31; C as for the f32 MFMA.  A lane's 8 accumulator registers r = 8 t + q of
// out-block b are therefore exactly its 8 k-slots of k-block kb = 2 b + t of the next layer: slot 8 h + q of
// k-block kb <-> k-pair p = 8 kb + q of the f32 kernel (cfeat(p, h) / pefeat(p, h, L)).
// Every f32 weight w is split by truncation into three bf16 planes, w = hi + mid + lo exactly.  The stream is cut
// into SLOTS of 48 KiB = 48 tiles of 1 KiB; tile (k-block, plane, out-block) = 64 lanes x 8 bf16:
//   slot[(((kb_in_slot * 3 + plane) * OB + ob) * 64 + lane) * 8 + q] = plane(W[32 ob + (lane & 31)][col(8 kb + q, lane >> 5)])
// a slot holds 16 / OB k-blocks (OB in {8, 4, 2, 1}); segments in consumption order:
//   mlps.0 (4 kb) | L1..L4 (16 kb each) | L5 h part (16) | L5 pe part (4) | L6 | L7 | rgb hidden' (16, OB 4) | dirs (2, OB 4) |
//   ins hidden' (16, OB 4) | ins_linear (8, OB 1/2/4)          (' = feature linear folded in, weights.py::fuse_heads)
// The table (biases, VALU heads) is the f32 blob's, unchanged.
constexpr int SPLIT_SLOT_WORDS = 12288;          // 48 KiB
constexpr int SPLIT_TILES_PER_SLOT = 48;
DMN_HD constexpr int split_kb_per_slot(int ob) { return 16 / ob; }
DMN_HD constexpr int split_slots(int nkb, int ob) { return (nkb + split_kb_per_slot(ob) - 1) / split_kb_per_slot(ob); }
DMN_HD constexpr int split_ob_ins(int obi) { return obi == 3 ? 4 : obi; }
struct SplitLayout {
    int C, OBI, OBX;              // logits, logit blocks, out-blocks used for the ins_linear segment (1, 2 or 4)
    int s_w0, s_trunk, s_l5pe, s_l6, s_rgbh, s_dirs, s_insh, s_inso, n_slots;   // first slot of each segment
    int64_t stream, total;        // word offsets: stream start (= TAB_FLOATS), total words (incl. 2 landing slots)
};
DMN_HD inline SplitLayout make_split_layout(int ins_num) {
    SplitLayout S{};
    S.C = ins_num + 1; S.OBI = (S.C + 31) / 32; S.OBX = split_ob_ins(S.OBI);
    int o = 0;
    S.s_w0 = o; o += split_slots(4, 8);
    S.s_trunk = o; o += 5 * split_slots(16, 8);       // L1..L4, L5 (h columns)
    S.s_l5pe = o; o += split_slots(4, 8);
    S.s_l6 = o; o += 2 * split_slots(16, 8);          // L6, L7
    S.s_rgbh = o; o += split_slots(16, 4);
    S.s_dirs = o; o += split_slots(2, 4);
    S.s_insh = o; o += split_slots(16, 4);
    S.s_inso = o; o += split_slots(8, S.OBX);
    S.n_slots = o;
    S.stream = TAB_FLOATS;
    S.total = S.stream + (int64_t)(o + 2) * SPLIT_SLOT_WORDS;
    return S;
}

// Split-bf16 blob of the OPT-IN data-gradient kernel (mlp_bwd_split.hip): W^T segments in the SplitLayout tile format,
//   slot[(((kb_in_slot * 3 + plane) * OB + ob) * 64 + lane) * 8 + q] = plane(W[out = cfeat(8 kb + q, lane >> 5)][in = 32 ob + (lane & 31)])
// ins_linear^T (2 OBI k-blocks, OB 4) | F^T (8 k-blocks, OB 8) | mlps.7^T .. mlps.1^T (16 k-blocks, OB 8 each; mlps.5: its h columns);
// the table in front is the f32 backward blob's (TAB_T_FLOATS: the two VALU heads).
struct SplitTLayout {
    int C, OBI;
    int s_inso, s_rgbf, s_stage, n_slots;
    int64_t stream, total;        // word offsets (stream = TAB_T_FLOATS), total incl. 2 landing slots
};
DMN_HD inline SplitTLayout make_split_layout_t(int ins_num) {
    SplitTLayout S{};
    S.C = ins_num + 1; S.OBI = (S.C + 31) / 32;
    int o = 0;
    S.s_inso = o; o += split_slots(2 * S.OBI, 4);
    S.s_rgbf = o; o += split_slots(8, 8);
    S.s_stage = o; o += 7 * split_slots(16, 8);
    S.n_slots = o;
    S.stream = 1024;
    S.total = S.stream + (int64_t)(o + 2) * SPLIT_SLOT_WORDS;
    return S;
}

// ---- split-f16 ("f16x2") blobs (opt-in; mlp_f16_impl.h, mlp_bwd_f16.hip) ------------------------------------------
// Every f32 weight w is split into TWO f16 planes, hi = f16(w), lo = f16(w - hi) (round to nearest; the lo plane lives in
// the f16 subnormal range for |w| < 2^-4, which v_mfma_f32_32x32x16_f16 honours: scripts/micro/mfma_f16.hip); a product is
// w_hi x_hi + w_hi x_lo + w_lo x_hi -- three MFMAs.  The stream is a flat sequence of GROUPS of 16 KiB:
//   group = 8 hi tiles | 8 lo tiles,  tile i <-> (k-block kb0 + i / nob, out-block ob0 + i % nob),  tile = 64 lanes x 8 f16:
//   group[(plane * 8 + i) * 512 + lane * 8 + q] = plane(W[32 ob + (lane & 31)][col(8 kb + q, lane >> 5)])
// consumed in order by a ring of F16_RING group slots in LDS, F16_LA groups ahead.  The kernels walk the network OUT-BLOCK-
// OUTER: a "pass" accumulates nob out-blocks over ALL their k-blocks (nob = 2 in the trunk: 4 passes x 4 groups per 256 -> 256
// layer), so that the ReLU + split epilogue of one pass rides in the MFMA gaps of the next.
// Forward order (fused heads, weights.py::fuse_heads):
//   mlps.0 (4 passes x 1 group, PE k-order) | mlps.1..4 | mlps.5 (per pass: 4 groups h + 1 group pts) | mlps.6 | mlps.7 |
//   rgb hidden' (nob 4: 8 groups h + 1 group dirs) | ins hidden' (nob 4: 8 groups) | rgb_linear (nob 1: 1 group) |
//   density (nob 1: 2 groups) | ins_linear (nob OBX: OBX groups)                                      = 140 + OBX groups
// Table (f32, TAB_FLOATS): every bias in accumulator order (bias_floats), in the order the passes need them; the bias of
// mlps.0 is a stream column instead (the pad slot of the position encoding, which the kernel feeds with 1.0).
constexpr int F16_GROUP_WORDS = 4096;            // 16 KiB
constexpr int F16_RING = 8;                      // LDS ring: 8 group slots = 128 KiB
constexpr int F16_LA = 6;                        // groups fetched ahead of the one being consumed
constexpr int F16_TAB_RGBH = 8 * 256, F16_TAB_INSH = F16_TAB_RGBH + 128, F16_TAB_DEN = F16_TAB_INSH + 128,
              F16_TAB_RGBO = F16_TAB_DEN + 32, F16_TAB_INSO = F16_TAB_RGBO + 32;            // + 32 OBX <= 2496 <= TAB_FLOATS
struct F16Layout {
    int C, OBI, OBX;
    int n_groups;                 // groups of the forward stream
    int64_t stream, total;        // word offsets: stream start (= TAB_FLOATS), total words incl. F16_LA landing groups
};
DMN_HD inline F16Layout make_f16_layout(int ins_num) {
    F16Layout S{};
    S.C = ins_num + 1; S.OBI = (S.C + 31) / 32; S.OBX = split_ob_ins(S.OBI);
    S.n_groups = 4 + 4 * 16 + 20 + 2 * 16 + 9 + 8 + 2 + 1 + S.OBX;
    S.stream = TAB_FLOATS;
    S.total = S.stream + (int64_t)(S.n_groups + F16_LA) * F16_GROUP_WORDS;
    return S;
}

// Split-f16 blob of the OPT-IN data-gradient kernel (mlp_bwd_f16.hip): W^T groups in the F16Layout group format
// (A tile rows = the layer's INPUTS 32 ob + (lane & 31), k = its OUTPUTS in accumulator order cfeat(8 kb + q, lane >> 5)):
//   ins_linear^T (nob 4: the 128 g2 rows; 2 OBI k-blocks = OBI groups) | F^T (4 passes x 2 groups, nob 2: 8 k-blocks = dg1) |
//   mlps.7^T .. mlps.1^T (4 passes x 4 groups each; mlps.5: its h columns)                             = 120 + OBI groups
// the table in front is the f32 backward blob's (TAB_T_FLOATS: the two VALU heads).
struct F16TLayout {
    int C, OBI, n_groups;
    int64_t stream, total;        // word offsets (stream = TAB_T_FLOATS), total incl. F16_LA landing groups
};
DMN_HD inline F16TLayout make_f16_layout_t(int ins_num) {
    F16TLayout S{};
    S.C = ins_num + 1; S.OBI = (S.C + 31) / 32;
    S.n_groups = S.OBI + 8 + 7 * 16;
    S.stream = 1024;
    S.total = S.stream + (int64_t)(S.n_groups + F16_LA) * F16_GROUP_WORDS;
    return S;
}

// Backward (dgrad) blob: the same segments with W^T as the A operand, dx^T = W^T . dy^T.
//   seg[((g*OB + ob)*64 + lane)*4 + kk] = W[out = cfeat(4g+kk, lane>>5)][in = ob*32 + (lane&31)]
//
// The two feature linears have no activation (dm_nerf.py:89,96), so the backward never materialises their
// gradients (heads.hip):  with A = rgb_feature_linears.0.weight[:, :256] and F = A . W_rgb_feature (128 x 256, formed
// once per weight update by head_product_kernel and appended to the flat parameter vector),
//     d h_7 (rgb branch) = W_rf^T (A^T dg1) = F^T dg1        -- ONE 128 -> 256 GEMM instead of 128 -> 256 -> 256,
// and the ins branch, whose input is h.detach() (:95), sends nothing to h_7 at all.  The weight gradients of the four
// linears follow from G = dg1 . h_7^T and Q = dg2 . h_7^T (wgrad.hip / heads.hip::head_unfuse_kernel).
constexpr int NSTAGE_T = 7;   // mlps.7^T .. mlps.1^T (mlps.5: its 256 h-columns)
constexpr int TAB_T_FLOATS = 1024;       // table of the backward blob: the two VALU heads (4 KiB)
constexpr int HEAD_F_FLOATS = HW * W;    // F = A . W_rgb_feature, row-major [128][256], stored behind the flat parameters
// Backward blob = [table: w_rgbo [c][half][64] | w_den [half][128]] [stream: ins_linear^T (1 padded quarter) |
// F^T (2) | 7 stages x 4] [2 dummy quarters].
struct BlobTLayout {
    int C, OBI;
    int64_t w_rgbo, w_den;   // table
    int64_t stream;
    int64_t t_inso;     // ins_linear^T:  K = 32*OBI logits (NKG = 4*OBI), rows 128 (OB = 4)
    int64_t t_rgbf;     // F^T:           K = 128 (NKG = 16), rows 256 = the h_7 features (OB = 8)
    int64_t t_stage;    // NSTAGE_T x (NKG = 32, OB = 8)
    int64_t total;
};
DMN_HD inline BlobTLayout make_layout_t(int ins_num) {
    BlobTLayout L;
    L.C = ins_num + 1;
    L.OBI = (L.C + 31) / 32;
    L.w_rgbo = 0;
    L.w_den = 3 * 2 * 64;
    int64_t o = TAB_T_FLOATS;
    L.stream = o;
    L.t_inso = o; o += QUARTER_FLOATS;
    L.t_rgbf = o; o += seg_floats(16, 8);
    L.t_stage = o; o += NSTAGE_T * seg_floats(32, 8);
    o += 2 * QUARTER_FLOATS;
    L.total = o;
    return L;
}

// Training workspace: activations saved by the forward for the backward, every tensor stored
// feature-major [rows][M] (row = feature in reference order, column = sample).
// 1-bit ReLU masks: per 32-sample block [8 layers][64 lanes][4 words] + [g1][64][2] + [g2][64][2] words;
// word (p >> 5) bit (p & 31) of a lane = (activation register p of that lane > 0), p = 16 b + r.
constexpr int BITS_WORDS_PER_BLOCK = 8 * 64 * 4 + 2 * 64 * 2;          // 2304 = 72 per sample
constexpr int SAVE_ROWS = POS_CH + DIR_CH + 8 * W + HW + HW + BITS_WORDS_PER_BLOCK / 32;   // 2466
struct SaveLayout {
    int64_t pe, de;      // embed(pts) [63][M], embed(viewdirs) [27][M]
    int64_t h;           // relu outputs of mlps.0..7: [8][256][M]
    // (rgb_feature / ins_feature are NOT saved: their weight gradients come from G = dg1 . h_7^T, Q = dg2 . h_7^T)
    int64_t g1, g2;      // relu outputs of rgb_feature_linears.0 / ins_feature_linears.0: [128][M] each.  In the GRADIENT workspace (same
                         // offsets) the two regions are ONE block-major tensor of 256 rows, dg1 = rows 0..127, dg2 = rows 128..255 of every
                         // 32-sample block, so that the weight-gradient kernel streams h_7 once against [dg1 ; dg2] (G and Q of heads.hip)
    int64_t bits;        // ReLU bit masks, BITS_WORDS_PER_BLOCK words per block (forward workspace only)
    int64_t total;
};
// Row length: M rounded up to a multiple of 32 so that a wave's 32-sample block never straddles
// the end of a row (tail lanes store into / read from the padding columns).
// Accumulator-layout training tensors keep, inside every group of 8 features, the memory row order
// 0,4,1,5,2,6,3,7 (mlp_common.h: TID-addressed stores).  Feature held by memory row rho:
DMN_HD constexpr int row_feature(int rho) { return (rho & ~7) | ((rho & 7) >> 1) | ((rho & 1) << 2); }

DMN_HD constexpr int64_t save_row_len(int64_t M) { return (M + 31) & ~(int64_t)31; }

DMN_HD inline SaveLayout make_save_layout(int64_t M_samples) {
    SaveLayout s;
    const int64_t M = save_row_len(M_samples);
    int64_t o = 0;
    s.pe = o; o += POS_CH * M;
    s.de = o; o += DIR_CH * M;
    s.h = o; o += 8 * (int64_t)W * M;
    s.g1 = o; o += (int64_t)HW * M;
    s.g2 = o; o += (int64_t)HW * M;
    s.bits = o; o += (int64_t)(BITS_WORDS_PER_BLOCK / 32) * M;
    s.total = o;
    return s;
}

}  // namespace dmn
