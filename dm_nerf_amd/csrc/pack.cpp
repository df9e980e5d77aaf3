// pack.cpp -- host-side: reference state_dict order -> kernel blob gather index.  No GPU needed.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/dmnerf_hip.h"
#include "common.h"
#include "layout.h"
#include "params.h"

using namespace dmn;

namespace {

enum KMap { K_ACC, K_POS, K_DIR };

int kcol(KMap km, int p, int half) {
    switch (km) {
        case K_ACC: return cfeat(p, half);
        case K_POS: return pefeat(p, half, POS_L);
        case K_DIR: return pefeat(p, half, DIR_L);
    }
    return -1;
}

void fill_seg(int32_t* idx, int64_t off, const Lin& l, int nkg, int ob_n, KMap km, int col_off) {
    for (int g = 0; g < nkg; ++g)
        for (int ob = 0; ob < ob_n; ++ob)
            for (int lane = 0; lane < 64; ++lane)
                for (int kk = 0; kk < 4; ++kk) {
                    int col = kcol(km, 4 * g + kk, lane >> 5);
                    int64_t src = col < 0 ? -1 : l.w(ob * 32 + (lane & 31), col + col_off);
                    idx[off + (((int64_t)g * ob_n + ob) * 64 + lane) * 4 + kk] = (int32_t)src;
                }
}

// transposed segment for the backward: rows = inputs of `l`, k = outputs of `l`
void fill_seg_t(int32_t* idx, int64_t off, const Lin& l, int nkg, int ob_n) {
    for (int g = 0; g < nkg; ++g)
        for (int ob = 0; ob < ob_n; ++ob)
            for (int lane = 0; lane < 64; ++lane)
                for (int kk = 0; kk < 4; ++kk) {
                    const int out = cfeat(4 * g + kk, lane >> 5);
                    const int in = ob * 32 + (lane & 31);
                    idx[off + (((int64_t)g * ob_n + ob) * 64 + lane) * 4 + kk] = (int32_t)l.w(out, in);
                }
}

void fill_bias(int32_t* idx, int64_t off, const Lin& l, int ob_n) {
    for (int ob = 0; ob < ob_n; ++ob)
        for (int half = 0; half < 2; ++half)
            for (int r = 0; r < 16; ++r)
                idx[off + (ob * 2 + half) * 16 + r] = (int32_t)l.b(32 * ob + (r & 3) + 8 * (r >> 2) + 4 * half);
}

}  // namespace

extern "C" int64_t dmnerf_param_count(int ins_num) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return -1;
    return make_params(ins_num).total;
}

extern "C" int64_t dmnerf_blob_floats(int ins_num) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return -1;
    return make_layout(ins_num).total;
}

static int build_pack_index(int ins_num, int32_t* idx, int64_t n_idx, bool fused) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS)
        return dmn_fail(DMNERF_E_ARG, "build_pack_index: ins_num %d outside [1,%d]", ins_num, DMNERF_MAX_LOGITS - 1);
    const BlobLayout L = make_layout(ins_num, fused);
    if (!idx || n_idx != L.total)
        return dmn_fail(DMNERF_E_ARG, "build_pack_index: need %lld index slots, got %lld", (long long)L.total, (long long)n_idx);
    const Params P = make_params(ins_num);
    for (int64_t i = 0; i < L.total; ++i) idx[i] = -1;

    fill_seg(idx, L.w0, P.mlps[0], 8, 8, K_POS, 0);
    fill_bias(idx, L.b0, P.mlps[0], 8);
    // the nine 256->256 stages: L1..L4, L5 (h columns 0..255), L6, L7, rgb_feature, ins_feature
    // (fused: the last two are folded into rgb_hidden / ins_hidden by the caller and have no segment)
    const Lin* stage[NSTAGE] = {&P.mlps[1], &P.mlps[2], &P.mlps[3], &P.mlps[4], &P.mlps[5],
                                &P.mlps[6], &P.mlps[7], &P.rgb_feature, &P.ins_feature};
    for (int s = 0; s < (fused ? NSTAGE - 2 : NSTAGE); ++s) {
        fill_seg(idx, stage_off(L, s), *stage[s], 32, 8, K_ACC, 0);
        fill_bias(idx, L.b_stage + s * bias_floats(8), *stage[s], 8);
    }
    fill_seg(idx, L.w5pe, P.mlps[5], 8, 8, K_POS, W);            // skip concat [h, pts] (dm_nerf.py:87)
    fill_seg(idx, L.w_rgbh, P.rgb_hidden, 32, 4, K_ACC, 0);
    fill_seg(idx, L.w_rgbh_dir, P.rgb_hidden, 4, 4, K_DIR, W);   // cat[rgb_feature, dirs] (dm_nerf.py:90)
    fill_bias(idx, L.b_rgbh, P.rgb_hidden, 4);
    fill_seg(idx, L.w_insh, P.ins_hidden, 32, 4, K_ACC, 0);
    fill_bias(idx, L.b_insh, P.ins_hidden, 4);
    fill_seg(idx, L.w_inso, P.ins_out, 16, L.OBI, K_ACC, 0);
    fill_bias(idx, L.b_inso, P.ins_out, L.OBI);
    for (int half = 0; half < 2; ++half)
        for (int p = 0; p < 128; ++p) idx[L.w_den + half * 128 + p] = (int32_t)P.density.w(0, cfeat(p, half));
    idx[L.b_den] = (int32_t)P.density.b(0);
    for (int c = 0; c < 3; ++c)
        for (int half = 0; half < 2; ++half)
            for (int p = 0; p < 64; ++p)
                idx[L.w_rgbo + (c * 2 + half) * 64 + p] = (int32_t)P.rgb_out.w(c, cfeat(p, half));
    for (int c = 0; c < 3; ++c) idx[L.b_rgbo + c] = (int32_t)P.rgb_out.b(c);
    return DMNERF_OK;
}

extern "C" int dmnerf_build_pack_index(int ins_num, int32_t* idx, int64_t n_idx) { return build_pack_index(ins_num, idx, n_idx, false); }

extern "C" int64_t dmnerf_blob_fused_floats(int ins_num) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return -1;
    return make_layout(ins_num, true).total;
}
extern "C" int dmnerf_build_pack_index_fused(int ins_num, int32_t* idx, int64_t n_idx) { return build_pack_index(ins_num, idx, n_idx, true); }

// ---- split-bf16 blob (layout.h::SplitLayout): one int32 per bf16 element = source parameter | plane << 28, or -1
static void fill_split_seg(int32_t* idx, int slot0, const Lin& l, int nkb, int ob_n, KMap km, int col_off) {
    const int kps = split_kb_per_slot(ob_n);
    for (int kb = 0; kb < nkb; ++kb)
        for (int plane = 0; plane < 3; ++plane)
            for (int ob = 0; ob < ob_n; ++ob)
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 8; ++q) {
                        const int col = kcol(km, 8 * kb + q, lane >> 5);
                        const int64_t src = col < 0 ? -1 : l.w(ob * 32 + (lane & 31), col + col_off);
                        const int64_t e = ((int64_t)(slot0 + kb / kps) * SPLIT_TILES_PER_SLOT + ((kb % kps) * 3 + plane) * ob_n + ob) * 512 + lane * 8 + q;
                        idx[e] = src < 0 ? -1 : (int32_t)(src | ((int64_t)plane << 28));
                    }
}

extern "C" int64_t dmnerf_blob_split_words(int ins_num) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return -1;
    return make_split_layout(ins_num).total;
}

extern "C" int dmnerf_build_pack_index_split(int ins_num, int32_t* idx, int64_t n_idx) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "build_pack_index_split: ins_num %d unsupported", ins_num);
    const SplitLayout S = make_split_layout(ins_num);
    const int64_t need = (S.total - S.stream) * 2;                  // bf16 elements of the stream (incl. landing slots)
    if (!idx || n_idx != need) return dmn_fail(DMNERF_E_ARG, "build_pack_index_split: need %lld index slots, got %lld", (long long)need, (long long)n_idx);
    const Params P = make_params(ins_num);
    for (int64_t i = 0; i < need; ++i) idx[i] = -1;
    fill_split_seg(idx, S.s_w0, P.mlps[0], 4, 8, K_POS, 0);
    const Lin* trunk[5] = {&P.mlps[1], &P.mlps[2], &P.mlps[3], &P.mlps[4], &P.mlps[5]};
    for (int s = 0; s < 5; ++s) fill_split_seg(idx, S.s_trunk + s * split_slots(16, 8), *trunk[s], 16, 8, K_ACC, 0);
    fill_split_seg(idx, S.s_l5pe, P.mlps[5], 4, 8, K_POS, W);
    fill_split_seg(idx, S.s_l6, P.mlps[6], 16, 8, K_ACC, 0);
    fill_split_seg(idx, S.s_l6 + split_slots(16, 8), P.mlps[7], 16, 8, K_ACC, 0);
    fill_split_seg(idx, S.s_rgbh, P.rgb_hidden, 16, 4, K_ACC, 0);      // (fused with rgb_feature_linear by the caller)
    fill_split_seg(idx, S.s_dirs, P.rgb_hidden, 2, 4, K_DIR, W);
    fill_split_seg(idx, S.s_insh, P.ins_hidden, 16, 4, K_ACC, 0);      // (fused with ins_feature_linear)
    fill_split_seg(idx, S.s_inso, P.ins_out, 8, S.OBX, K_ACC, 0);
    return DMNERF_OK;
}

// transposed split segment: k = outputs of `l` (accumulator order), rows = inputs of `l`
static void fill_split_seg_t(int32_t* idx, int slot0, const Lin& l, int nkb, int ob_n) {
    const int kps = split_kb_per_slot(ob_n);
    for (int kb = 0; kb < nkb; ++kb)
        for (int plane = 0; plane < 3; ++plane)
            for (int ob = 0; ob < ob_n; ++ob)
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 8; ++q) {
                        const int64_t src = l.w(cfeat(8 * kb + q, lane >> 5), ob * 32 + (lane & 31));
                        const int64_t e = ((int64_t)(slot0 + kb / kps) * SPLIT_TILES_PER_SLOT + ((kb % kps) * 3 + plane) * ob_n + ob) * 512 + lane * 8 + q;
                        idx[e] = src < 0 ? -1 : (int32_t)(src | ((int64_t)plane << 28));
                    }
}

extern "C" int64_t dmnerf_blob_t_split_words(int ins_num) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return -1;
    return make_split_layout_t(ins_num).total;
}

// sources: [flat parameters | F] like dmnerf_build_pack_index_t (F = head product, behind the parameters)
extern "C" int dmnerf_build_pack_index_t_split(int ins_num, int32_t* idx, int64_t n_idx) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "build_pack_index_t_split: ins_num %d unsupported", ins_num);
    const SplitTLayout S = make_split_layout_t(ins_num);
    const int64_t need = (S.total - S.stream) * 2;
    if (!idx || n_idx != need) return dmn_fail(DMNERF_E_ARG, "build_pack_index_t_split: need %lld index slots, got %lld", (long long)need, (long long)n_idx);
    const Params P = make_params(ins_num);
    for (int64_t i = 0; i < need; ++i) idx[i] = -1;
    fill_split_seg_t(idx, S.s_inso, P.ins_out, 2 * S.OBI, 4);
    Lin F;
    F.w_off = P.total; F.b_off = -1; F.out = HW; F.in = W;
    fill_split_seg_t(idx, S.s_rgbf, F, 8, 8);
    const Lin* stage[7] = {&P.mlps[7], &P.mlps[6], &P.mlps[5], &P.mlps[4], &P.mlps[3], &P.mlps[2], &P.mlps[1]};
    for (int s = 0; s < 7; ++s) fill_split_seg_t(idx, S.s_stage + s * split_slots(16, 8), *stage[s], 16, 8);
    return DMNERF_OK;
}

// ---- split-f16 blob (layout.h::F16Layout): table index (one int32 per float) + stream index (one int32 per f16 element =
// source parameter | plane << 28, or -1), groups in the order mlp_f16_impl.h consumes them
namespace {
struct F16Group { const Lin* l; int nob, ob0, kb0; KMap km; int col_off; bool bias_pad; };   // bias_pad: the PE pad slot's column holds the bias

void f16_forward_groups(const Params& P, int obx, std::vector<F16Group>& g) {
    for (int p = 0; p < 4; ++p) g.push_back({&P.mlps[0], 2, 2 * p, 0, K_POS, 0, true});     // (the kernel feeds 1.0 in the pad slot)
    for (int l = 1; l < 8; ++l)
        for (int p = 0; p < 4; ++p) {
            for (int q = 0; q < 4; ++q) g.push_back({&P.mlps[l], 2, 2 * p, 4 * q, K_ACC, 0, false});
            if (l == 5) g.push_back({&P.mlps[5], 2, 2 * p, 0, K_POS, W, false});           // skip concat [h, pts] (dm_nerf.py:87)
        }
    for (int q = 0; q < 8; ++q) g.push_back({&P.rgb_hidden, 4, 0, 2 * q, K_ACC, 0, false});  // (fused with rgb_feature_linear by the caller)
    g.push_back({&P.rgb_hidden, 4, 0, 0, K_DIR, W, false});                                // cat[rgb_feature, dirs] (dm_nerf.py:90)
    for (int q = 0; q < 8; ++q) g.push_back({&P.ins_hidden, 4, 0, 2 * q, K_ACC, 0, false});  // (fused with ins_feature_linear)
    g.push_back({&P.rgb_out, 1, 0, 0, K_ACC, 0, false});
    for (int q = 0; q < 2; ++q) g.push_back({&P.density, 1, 0, 8 * q, K_ACC, 0, false});
    for (int q = 0; q < obx; ++q) g.push_back({&P.ins_out, obx, 0, q * (8 / obx), K_ACC, 0, false});
}

void fill_f16_groups(int32_t* idx, const std::vector<F16Group>& groups) {
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const F16Group& G = groups[gi];
        for (int plane = 0; plane < 2; ++plane)
            for (int i = 0; i < 8; ++i) {
                const int kb = G.kb0 + i / G.nob, ob = G.ob0 + i % G.nob;
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 8; ++q) {
                        const int col = kcol(G.km, 8 * kb + q, lane >> 5);
                        int64_t src = col < 0 ? -1 : G.l->w(ob * 32 + (lane & 31), col + G.col_off);
                        if (G.bias_pad && 8 * kb + q == 1 && (lane >> 5) == 1) src = G.l->b(ob * 32 + (lane & 31));
                        idx[((int64_t)gi * 16 + plane * 8 + i) * 512 + lane * 8 + q] = src < 0 ? -1 : (int32_t)(src | ((int64_t)plane << 28));
                    }
            }
    }
}
}  // namespace

extern "C" int64_t dmnerf_blob_f16_words(int ins_num) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return -1;
    return make_f16_layout(ins_num).total;
}

// idx_tab [TAB_FLOATS]: float gather for the bias table; idx_stream [(total - TAB_FLOATS) * 2]: the f16 elements
extern "C" int dmnerf_build_pack_index_f16(int ins_num, int32_t* idx_tab, int64_t n_tab, int32_t* idx_stream, int64_t n_stream) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "build_pack_index_f16: ins_num %d unsupported", ins_num);
    const F16Layout S = make_f16_layout(ins_num);
    const int64_t need = (S.total - S.stream) * 2;
    if (!idx_tab || !idx_stream || n_tab != TAB_FLOATS || n_stream != need)
        return dmn_fail(DMNERF_E_ARG, "build_pack_index_f16: need %d + %lld index slots, got %lld + %lld", TAB_FLOATS, (long long)need, (long long)n_tab, (long long)n_stream);
    const Params P = make_params(ins_num);
    for (int64_t i = 0; i < TAB_FLOATS; ++i) idx_tab[i] = -1;
    for (int64_t i = 0; i < need; ++i) idx_stream[i] = -1;
    for (int l = 1; l < 8; ++l) fill_bias(idx_tab, l * 256, P.mlps[l], 8);          // (mlps.0: its bias is a stream column; zeros here)
    fill_bias(idx_tab, F16_TAB_RGBH, P.rgb_hidden, 4);
    fill_bias(idx_tab, F16_TAB_INSH, P.ins_hidden, 4);
    fill_bias(idx_tab, F16_TAB_DEN, P.density, 1);
    fill_bias(idx_tab, F16_TAB_RGBO, P.rgb_out, 1);
    fill_bias(idx_tab, F16_TAB_INSO, P.ins_out, S.OBX);
    std::vector<F16Group> groups;
    f16_forward_groups(P, S.OBX, groups);
    if ((int)groups.size() != S.n_groups) return dmn_fail(DMNERF_E_ARG, "build_pack_index_f16: internal group count %d != %d", (int)groups.size(), S.n_groups);
    fill_f16_groups(idx_stream, groups);
    return DMNERF_OK;
}

// ---- split-f16 W^T blob (layout.h::F16TLayout); sources: [flat parameters | F] like dmnerf_build_pack_index_t
extern "C" int64_t dmnerf_blob_t_f16_words(int ins_num) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return -1;
    return make_f16_layout_t(ins_num).total;
}

extern "C" int dmnerf_build_pack_index_t_f16(int ins_num, int32_t* idx, int64_t n_idx) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return dmn_fail(DMNERF_E_ARG, "build_pack_index_t_f16: ins_num %d unsupported", ins_num);
    const F16TLayout S = make_f16_layout_t(ins_num);
    const int64_t need = (S.total - S.stream) * 2;
    if (!idx || n_idx != need) return dmn_fail(DMNERF_E_ARG, "build_pack_index_t_f16: need %lld index slots, got %lld", (long long)need, (long long)n_idx);
    const Params P = make_params(ins_num);
    for (int64_t i = 0; i < need; ++i) idx[i] = -1;
    Lin F;
    F.w_off = P.total; F.b_off = -1; F.out = HW; F.in = W;
    struct TG { const Lin* l; int nob, ob0, kb0; };
    std::vector<TG> groups;
    for (int q = 0; q < S.OBI; ++q) groups.push_back({&P.ins_out, 4, 0, 2 * q});
    for (int p = 0; p < 4; ++p)
        for (int q = 0; q < 2; ++q) groups.push_back({&F, 2, 2 * p, 4 * q});
    const Lin* stage[7] = {&P.mlps[7], &P.mlps[6], &P.mlps[5], &P.mlps[4], &P.mlps[3], &P.mlps[2], &P.mlps[1]};
    for (int s = 0; s < 7; ++s)
        for (int p = 0; p < 4; ++p)
            for (int q = 0; q < 4; ++q) groups.push_back({stage[s], 2, 2 * p, 4 * q});
    if ((int)groups.size() != S.n_groups) return dmn_fail(DMNERF_E_ARG, "build_pack_index_t_f16: internal group count");
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const TG& G = groups[gi];
        for (int plane = 0; plane < 2; ++plane)
            for (int i = 0; i < 8; ++i) {
                const int kb = G.kb0 + i / G.nob, ob = G.ob0 + i % G.nob;
                for (int lane = 0; lane < 64; ++lane)
                    for (int q = 0; q < 8; ++q) {
                        const int64_t src = G.l->w(cfeat(8 * kb + q, lane >> 5), ob * 32 + (lane & 31));
                        idx[((int64_t)gi * 16 + plane * 8 + i) * 512 + lane * 8 + q] = src < 0 ? -1 : (int32_t)(src | ((int64_t)plane << 28));
                    }
            }
    }
    return DMNERF_OK;
}

extern "C" int64_t dmnerf_blob_t_floats(int ins_num) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS) return -1;
    return make_layout_t(ins_num).total;
}

extern "C" int dmnerf_build_pack_index_t(int ins_num, int32_t* idx, int64_t n_idx) {
    if (ins_num < 1 || ins_num + 1 > DMNERF_MAX_LOGITS)
        return dmn_fail(DMNERF_E_ARG, "build_pack_index_t: ins_num %d outside [1,%d]", ins_num, DMNERF_MAX_LOGITS - 1);
    const BlobTLayout L = make_layout_t(ins_num);
    if (!idx || n_idx != L.total)
        return dmn_fail(DMNERF_E_ARG, "build_pack_index_t: need %lld index slots, got %lld", (long long)L.total, (long long)n_idx);
    const Params P = make_params(ins_num);
    for (int64_t i = 0; i < L.total; ++i) idx[i] = -1;
    fill_seg_t(idx, L.t_inso, P.ins_out, 4 * L.OBI, 4);
    // F = rgb_feature_linears.0.weight[:, :256] . rgb_feature_linear.weight: a [128][256] matrix that
    // dmnerf_head_product writes BEHIND the flat parameters (indices >= param_count); as a "layer" it maps the 256 h_7
    // features to the 128 hidden units, and its transpose carries dg1 back to h_7 in one GEMM (layout.h)
    Lin F;
    F.w_off = P.total; F.b_off = -1; F.out = HW; F.in = W;
    fill_seg_t(idx, L.t_rgbf, F, 16, 8);
    const Lin* stage[NSTAGE_T] = {&P.mlps[7], &P.mlps[6], &P.mlps[5], &P.mlps[4], &P.mlps[3], &P.mlps[2], &P.mlps[1]};
    for (int s = 0; s < NSTAGE_T; ++s) fill_seg_t(idx, L.t_stage + s * seg_floats(32, 8), *stage[s], 32, 8);
    // table: the VALU heads in accumulator order (same packing as the forward table)
    for (int c = 0; c < 3; ++c)
        for (int half = 0; half < 2; ++half)
            for (int q = 0; q < 64; ++q)
                idx[L.w_rgbo + (c * 2 + half) * 64 + q] = (int32_t)P.rgb_out.w(c, cfeat(q, half));
    for (int half = 0; half < 2; ++half)
        for (int q = 0; q < 128; ++q) idx[L.w_den + half * 128 + q] = (int32_t)P.density.w(0, cfeat(q, half));
    return DMNERF_OK;
}
