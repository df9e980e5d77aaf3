"""Host-side checks (no GPU) of the two opt-in inference blobs: the fused-heads f32 blob and the split-bf16 blob
are re-arrangements of the same parameters as the default blob (include/dmnerf_hip.h; csrc/layout.h, pack.cpp)."""
import ctypes

import numpy as np
import pytest

from dm_nerf_amd import _lib, weights as W

TAB, QUARTER, SLOT_ELEMS = 4096, 16384, 12288 * 2


def cfeat(p, h):
    return 32 * (p >> 4) + ((p & 15) & 3) + 8 * ((p & 15) >> 2) + 4 * h


@pytest.mark.parametrize("ins_num", [13, 59, 93])
def test_fused_index_is_the_default_index_without_the_two_feature_stages(ins_num):
    a, b = W.pack_index_host(ins_num), W.pack_index_fused_host(ins_num)
    assert len(a) - len(b) == 8 * QUARTER                       # rgb_feature (st7) and ins_feature (st8): 4 quarters each
    upto_st6 = TAB + QUARTER * (1 + 20 + 1 + 8)                 # w0 | st0..st4 | w5pe | st5 st6
    assert np.array_equal(a[TAB:upto_st6], b[TAB:upto_st6])
    # table: identical except the (absent) biases of stages 7 and 8
    assert np.array_equal(np.delete(a[:TAB], np.s_[2048:2560]), np.delete(b[:TAB], np.s_[2048:2560]))
    assert (b[2048:2560] == -1).all()
    # rgb hidden (2 quarters) + dirs (1) directly follow st6; ins hidden (2) + ins_linear (1) follow them
    sa, sb = upto_st6 + 4 * QUARTER, upto_st6
    assert np.array_equal(a[sa:sa + 3 * QUARTER], b[sb:sb + 3 * QUARTER])
    sa, sb = sa + 3 * QUARTER + 4 * QUARTER, sb + 3 * QUARTER
    assert np.array_equal(a[sa:sa + 3 * QUARTER], b[sb:sb + 3 * QUARTER])


@pytest.mark.parametrize("ins_num", [13, 59, 93])
def test_split_index_addresses_the_right_parameters(ins_num):
    lib = _lib.load()
    total = lib.dmnerf_blob_split_words(ins_num)
    n = (total - TAB) * 2
    idx = np.empty(n, dtype=np.int32)
    _lib.check(lib.dmnerf_build_pack_index_split(ins_num, idx.ctypes.data_as(ctypes.c_void_p), n), "split index")
    C = ins_num + 1
    obx = {1: 1, 2: 2, 3: 4, 4: 4}[(C + 31) // 32]
    # parameter offsets in the flat vector (reference state_dict order)
    shapes = {"mlps.0": (256, 63), **{f"mlps.{i}": (256, 319 if i == 5 else 256) for i in range(1, 8)},
              "rgb_feature_linear": (256, 256), "ins_feature_linear": (256, 256), "rgb_feature_linears.0": (128, 283),
              "ins_feature_linears.0": (128, 256), "density_linear": (1, 256), "ins_linear": (C, 128), "rgb_linear": (3, 128)}
    off, o = {}, 0
    for m in W.PARAM_MODULES:
        off[m] = o
        o += shapes[m][0] * shapes[m][1] + shapes[m][0]
    assert o == lib.dmnerf_param_count(ins_num)
    # slots: w0 2 | L1..L5h 5x8 | L5 pe 2 | L6, L7 2x8 | rgb hidden 4 | dirs 1 | ins hidden 4 | ins_linear 1-2 | 2 landing slots
    n_inso = -(-8 // (16 // obx))                                   # ins_linear: 8 k-blocks, 16 / OB per slot
    assert n == (2 + 40 + 2 + 16 + 4 + 1 + 4 + n_inso + 2) * SLOT_ELEMS
    s_l7, s_rgbh, s_insh, s_inso = 2 + 40 + 2 + 8, 2 + 40 + 2 + 16, 2 + 40 + 2 + 16 + 4 + 1, 2 + 40 + 2 + 16 + 4 + 1 + 4

    def check(slot0, mod, nkb, ob_n, ncols, col0=0):
        rows, ld = shapes[mod]
        kps = 16 // ob_n
        for kb in (0, 1, nkb // 2, nkb - 1):
            for plane in range(3):
                for ob in range(ob_n):
                    for lane in (0, 7, 31, 32, 63):
                        for q in range(8):
                            e = ((slot0 + kb // kps) * 48 + ((kb % kps) * 3 + plane) * ob_n + ob) * 512 + lane * 8 + q
                            row, col = ob * 32 + (lane & 31), cfeat(8 * kb + q, lane >> 5)
                            want = -1 if (row >= rows or col >= ncols) else (off[mod] + row * ld + col0 + col) | (plane << 28)
                            assert idx[e] == want, (mod, kb, plane, ob, lane, q)

    check(s_l7, "mlps.7", 16, 8, 256)
    check(s_rgbh, "rgb_feature_linears.0", 16, 4, 256)
    check(s_insh, "ins_feature_linears.0", 16, 4, 256)
    check(s_inso, "ins_linear", 8, obx, 128)
    assert (idx[(s_inso + n_inso) * SLOT_ELEMS:] == -1).all()      # landing slots are zero


@pytest.mark.parametrize("ins_num", [13, 59, 93, 120])
def test_transposed_split_index_addresses_the_right_parameters(ins_num):
    """The W^T split stream of the opt-in data-gradient kernel (layout.h::SplitTLayout): ins_linear^T | F^T | mlps.7^T .. mlps.1^T,
    k = the layer's OUTPUTS in accumulator order, rows = its inputs; F sits behind the flat parameters."""
    lib = _lib.load()
    TAB_T = 1024
    total = lib.dmnerf_blob_t_split_words(ins_num)
    n = (total - TAB_T) * 2
    idx = np.empty(n, dtype=np.int32)
    _lib.check(lib.dmnerf_build_pack_index_t_split(ins_num, idx.ctypes.data_as(ctypes.c_void_p), n), "split index (transposed)")
    C = ins_num + 1
    obi = (C + 31) // 32
    shapes = {"mlps.0": (256, 63), **{f"mlps.{i}": (256, 319 if i == 5 else 256) for i in range(1, 8)},
              "rgb_feature_linear": (256, 256), "ins_feature_linear": (256, 256), "rgb_feature_linears.0": (128, 283),
              "ins_feature_linears.0": (128, 256), "density_linear": (1, 256), "ins_linear": (C, 128), "rgb_linear": (3, 128)}
    off, o = {}, 0
    for m in W.PARAM_MODULES:
        off[m] = o
        o += shapes[m][0] * shapes[m][1] + shapes[m][0]
    off["F"], shapes["F"] = o, (128, 256)                          # head product [rgb hidden 128][h_7 256], behind the parameters
    n_inso = -(-2 * obi // 4)                                      # 2 OBI k-blocks, OB 4: 4 k-blocks per slot
    assert n == (n_inso + 4 + 7 * 8 + 2) * SLOT_ELEMS              # F^T: 8 k-blocks of OB 8 = 4 slots; a stage: 8 slots; 2 landing slots

    def check(slot0, mod, nkb, ob_n):
        outs, ld = shapes[mod]
        kps = 16 // ob_n
        for kb in sorted({0, 1, nkb // 2, nkb - 1}):
            for plane in range(3):
                for ob in range(ob_n):
                    for lane in (0, 7, 31, 32, 63):
                        for q in range(8):
                            e = ((slot0 + kb // kps) * 48 + ((kb % kps) * 3 + plane) * ob_n + ob) * 512 + lane * 8 + q
                            out, inp = cfeat(8 * kb + q, lane >> 5), ob * 32 + (lane & 31)
                            want = -1 if (out >= outs or inp >= min(ld, 32 * ob_n)) else (off[mod] + out * ld + inp) | (plane << 28)
                            assert idx[e] == want, (mod, kb, plane, ob, lane, q, idx[e], want)

    check(0, "ins_linear", 2 * obi, 4)
    check(n_inso, "F", 8, 8)
    for s, l in enumerate((7, 6, 5, 4, 3, 2, 1)):
        check(n_inso + 4 + 8 * s, f"mlps.{l}", 16, 8)             # (mlps.5: its first 256 input columns = h; the 63 pts columns get no dgrad)
    assert (idx[(n_inso + 4 + 56) * SLOT_ELEMS:] == -1).all()      # landing slots


@pytest.mark.parametrize("ins_num", [13, 59, 93, 120])
def test_f16_index_addresses_the_right_parameters(ins_num):
    """The split-f16 forward blob (layout.h::F16Layout, csrc/mlp_f16_impl.h): a flat stream of 16 KiB groups = 8 hi tiles + 8 lo
    tiles, tile i <-> (k-block kb0 + i / nob, out-block ob0 + i % nob), in the order the kernel's passes consume them; the bias
    of mlps.0 sits in the stream column of the position encoding's pad slot, every other bias in the table."""
    lib = _lib.load()
    total = lib.dmnerf_blob_f16_words(ins_num)
    C = ins_num + 1
    obx = {1: 1, 2: 2, 3: 4, 4: 4}[(C + 31) // 32]
    n_groups = 140 + obx
    assert total == TAB + (n_groups + 6) * 4096                    # + F16_LA landing groups
    n = (total - TAB) * 2
    tab, idx = np.empty(TAB, dtype=np.int32), np.empty(n, dtype=np.int32)
    _lib.check(lib.dmnerf_build_pack_index_f16(ins_num, tab.ctypes.data_as(ctypes.c_void_p), TAB, idx.ctypes.data_as(ctypes.c_void_p), n), "f16 index")
    shapes = {"mlps.0": (256, 63), **{f"mlps.{i}": (256, 319 if i == 5 else 256) for i in range(1, 8)},
              "rgb_feature_linear": (256, 256), "ins_feature_linear": (256, 256), "rgb_feature_linears.0": (128, 283),
              "ins_feature_linears.0": (128, 256), "density_linear": (1, 256), "ins_linear": (C, 128), "rgb_linear": (3, 128)}
    off, o = {}, 0
    for m in W.PARAM_MODULES:
        off[m] = o
        o += shapes[m][0] * shapes[m][1] + shapes[m][0]

    def pefeat(p, h, L):
        if p == 0:
            return h
        if p == 1:
            return -1 if h else 2
        q = p - 2
        return -1 if q >= 3 * L else 3 + 6 * (q // 3) + 3 * h + q % 3

    # the kernel's pass order: (module, nob, ob0, kb0, k-order, column offset, bias in the pad slot)
    groups = [("mlps.0", 2, 2 * p, 0, "pos", 0, True) for p in range(4)]
    for l in range(1, 8):
        for p in range(4):
            groups += [(f"mlps.{l}", 2, 2 * p, 4 * q, "acc", 0, False) for q in range(4)]
            if l == 5:
                groups.append(("mlps.5", 2, 2 * p, 0, "pos", 256, False))
    groups += [("rgb_feature_linears.0", 4, 0, 2 * q, "acc", 0, False) for q in range(8)] + [("rgb_feature_linears.0", 4, 0, 0, "dir", 256, False)]
    groups += [("ins_feature_linears.0", 4, 0, 2 * q, "acc", 0, False) for q in range(8)]
    groups += [("rgb_linear", 1, 0, 0, "acc", 0, False)] + [("density_linear", 1, 0, 8 * q, "acc", 0, False) for q in range(2)]
    groups += [("ins_linear", obx, 0, q * (8 // obx), "acc", 0, False) for q in range(obx)]
    assert len(groups) == n_groups
    rng = np.random.RandomState(ins_num)
    for gi in sorted(set(rng.randint(0, n_groups, 40)) | {0, 3, 4, 84, 88, 100, n_groups - 1}):
        mod, nob, ob0, kb0, order, col0, bias_pad = groups[gi]
        rows, ld = shapes[mod]
        for plane in range(2):
            for i in range(8):
                kb, ob = kb0 + i // nob, ob0 + i % nob
                for lane in (0, 5, 31, 32, 63):
                    for q in range(8):
                        p_, h = 8 * kb + q, lane >> 5
                        col = cfeat(p_, h) if order == "acc" else pefeat(p_, h, 10 if order == "pos" else 4)
                        row = ob * 32 + (lane & 31)
                        ncols = ld - col0 if order != "acc" else min(ld, 256 if mod.startswith("mlps") or "feature" in mod else ld)
                        if row >= rows or col < 0 or col >= ncols:
                            want = -1
                            if bias_pad and p_ == 1 and h == 1 and row < rows:
                                want = (off[mod] + rows * ld + row) | (plane << 28)          # the bias behind the weight matrix
                        else:
                            want = (off[mod] + row * ld + col0 + col) | (plane << 28)
                        e = (gi * 16 + plane * 8 + i) * 512 + lane * 8 + q
                        assert idx[e] == want, (gi, mod, plane, i, lane, q, idx[e], want)
    assert (idx[n_groups * 16 * 512:] == -1).all()                 # landing groups are zero
    # table: biases in accumulator order, pass order; mlps.0 has none (zeros)
    assert (tab[:256] == -1).all()
    for l in (1, 5, 7):
        for ob in (0, 7):
            for half in (0, 1):
                for r in (0, 5, 15):
                    assert tab[l * 256 + (ob * 2 + half) * 16 + r] == off[f"mlps.{l}"] + 256 * shapes[f"mlps.{l}"][1] + 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * half
    assert tab[2048] == off["rgb_feature_linears.0"] + 128 * 283 and tab[2048 + 128] == off["ins_feature_linears.0"] + 128 * 256
    assert tab[2304] == off["density_linear"] + 256 and (tab[2305:2336] == -1).all()
    assert [tab[2336 + r] for r in (0, 1, 2)] == [off["rgb_linear"] + 3 * 128 + c for c in (0, 1, 2)] and tab[2336 + 3] == -1
    assert tab[2368] == off["ins_linear"] + C * 128


def test_fuse_heads_is_the_same_function():
    import torch
    from oracle import ref_cpu as O
    sd = O.make_weights(9, 13, gain=1.7, sigma_bias=0.3)
    g = torch.Generator().manual_seed(9)
    x = torch.cat([O.embed(torch.randn(64, 3, generator=g), 10), O.embed(torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1), 4)], -1)
    want = O.mlp_forward(sd, x)
    sf = O.fuse_heads(sd)
    # evaluate the fused network: hidden layers take h directly
    xp, xv = x[:, :63], x[:, 63:]
    h = xp
    for i in range(8):
        h = torch.relu(h @ sf[f"mlps.{i}.weight"].t() + sf[f"mlps.{i}.bias"])
        if i == 4:
            h = torch.cat([h, xp], -1)
    den = h @ sf["density_linear.weight"].t() + sf["density_linear.bias"]
    hr = torch.relu(torch.cat([h, xv], -1) @ sf["rgb_feature_linears.0.weight"].t() + sf["rgb_feature_linears.0.bias"])
    rgb = hr @ sf["rgb_linear.weight"].t() + sf["rgb_linear.bias"]
    hi = torch.relu(h @ sf["ins_feature_linears.0.weight"].t() + sf["ins_feature_linears.0.bias"])
    ins = hi @ sf["ins_linear.weight"].t() + sf["ins_linear.bias"]
    got = torch.cat([rgb, den, ins], -1)
    assert float(((got - want).abs() / (1 + want.abs())).max()) <= 2e-6
