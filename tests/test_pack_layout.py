"""Host logic, no GPU: the weight-blob layout + the register dataflow of the fused MLP kernel.

A numpy model of v_mfma_f32_32x32x2_f32's lane<->element maps (MI355X guide: A[i=l&31][k=l>>5],
B[k=l>>5][j=l&31], C[j=l&31][i=(r&3)+8(r>>2)+4(l>>5)]) replays exactly the sequence of
loads / MFMAs mlp_fwd.hip issues for one wave, reading the packed blob, and must reproduce the
oracle MLP.  This pins pack.cpp + layout.h against the kernel's implicit k-order before any GPU run.
"""
import ctypes

import numpy as np
import torch

from dm_nerf_amd import _lib, weights
from oracle import ref_cpu as O

LANE = np.arange(64)
HALF = LANE >> 5


def crow(r, half):
    return (r & 3) + 8 * (r >> 2) + 4 * half


def mfma(a, b, c):
    """a, b: [64] lane registers; c: [64,16] accumulator registers.  Returns new c."""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    A[LANE & 31, HALF] = a
    B[HALF, LANE & 31] = b
    D = A @ B
    out = c.copy()
    for r in range(16):
        out[:, r] += D[crow(r, HALF), LANE & 31]
    return out


def layout(ins_num):
    """Python mirror of csrc/layout.h::make_layout: [table | weight stream in consumption order | 2 dummy quarters]."""
    C = ins_num + 1
    OBI = (C + 31) // 32
    Q = 16384
    L = {}
    o = 0
    def take(name, n):
        nonlocal o
        L[name] = o; o += n
    take("b0", 256); take("b_stage", 9 * 256); take("b_rgbh", 128); take("b_insh", 128); take("b_inso", OBI * 32)
    take("w_den", 256); take("b_den", 4); take("w_rgbo", 384); take("b_rgbo", 4)
    assert o <= 4096
    o = 4096
    take("w0", Q); take("w_stage_lo", 5 * 65536); take("w5pe", Q); take("w_stage_mid", 3 * 65536)
    take("w_rgbh", 32768); take("w_rgbh_dir", Q); take("w_stage_hi", 65536); take("w_insh", 32768); take("w_inso", Q)
    o += 2 * Q
    L["total"] = o; L["OBI"] = OBI; L["C"] = C
    return L


def stage_off(L, st):
    return L["w_stage_lo"] + st * 65536 if st < 5 else (L["w_stage_mid"] + (st - 5) * 65536 if st < 8 else L["w_stage_hi"])


def gemm_seg(blob, seg, nkg, ob_n, B, acc):
    """B: list of [64,16] register blocks (k-pair p = B[p>>4][:, p&15]); acc: list of [64,16]."""
    for g in range(nkg):
        a = [np.stack([blob[seg + ((g * ob_n + ob) * 64 + LANE) * 4 + kk] for kk in range(4)], 1) for ob in range(ob_n)]
        for kk in range(4):
            p = g * 4 + kk
            for ob in range(ob_n):
                acc[ob] = mfma(a[ob][:, kk], B[p >> 4][:, p & 15], acc[ob])
    return acc


def init_bias(blob, seg, ob_n):
    return [np.stack([blob[seg + (ob * 2 + HALF) * 16 + r] for r in range(16)], 1).astype(np.float64) for ob in range(ob_n)]


def encode_regs(e, L_freq, nv):
    """Register image of one encoded 3-vector per lane; e: [64, 3+6L] reference-order encoding."""
    out = [np.zeros((64, 16)) for _ in range(nv)]
    out[0][:, 0] = np.where(HALF == 1, e[:, 1], e[:, 0])
    out[0][:, 1] = np.where(HALF == 1, 0.0, e[:, 2])
    for k in range(L_freq):
        for c in range(3):
            p = 2 + 3 * k + c
            out[p >> 4][:, p & 15] = e[LANE, 3 + 6 * k + 3 * HALF + c]
    return out


def emulate_wave(blob, ins_num, x):
    """x: [32, 90] embedded rows of the wave's 32 samples -> raw [32, 4+C]."""
    L = layout(ins_num)
    xs = x[LANE & 31]
    pe = encode_regs(xs[:, :63], 10, 2)
    de = encode_regs(xs[:, 63:], 4, 1)
    relu = lambda blocks: [np.maximum(b, 0) for b in blocks]
    acc = gemm_seg(blob, L["w0"], 8, 8, pe, init_bias(blob, L["b0"], 8))
    h = relu(acc)
    raw = np.zeros((32, 4 + L["C"]))
    for st in range(9):
        acc = init_bias(blob, L["b_stage"] + st * 256, 8)
        acc = gemm_seg(blob, stage_off(L, st), 32, 8, h, acc)
        if st == 4:
            acc = gemm_seg(blob, L["w5pe"], 8, 8, pe, acc)
        if st < 7:
            h = relu(acc)
            if st == 6:
                part = sum(h[p >> 4][:, p & 15] * blob[L["w_den"] + HALF * 128 + p] for p in range(128))
                sigma = part + part[LANE ^ 32] + blob[L["b_den"]]
                raw[:, 3] = sigma[:32]
        elif st == 7:
            hid = gemm_seg(blob, L["w_rgbh"], 32, 4, acc, init_bias(blob, L["b_rgbh"], 4))
            hid = relu(gemm_seg(blob, L["w_rgbh_dir"], 4, 4, de, hid))
            for c in range(3):
                part = sum(hid[p >> 4][:, p & 15] * blob[L["w_rgbo"] + (c * 2 + HALF) * 64 + p] for p in range(64))
                raw[:, c] = (part + part[LANE ^ 32] + blob[L["b_rgbo"] + c])[:32]
        else:
            hid = relu(gemm_seg(blob, L["w_insh"], 32, 4, acc, init_bias(blob, L["b_insh"], 4)))
            io = gemm_seg(blob, L["w_inso"], 16, L["OBI"], hid, init_bias(blob, L["b_inso"], L["OBI"]))
            for b in range(L["OBI"]):
                for r in range(16):
                    for half in range(2):
                        ch = 32 * b + crow(r, half)
                        if ch < L["C"]:
                            raw[:, 4 + ch] = io[b][half * 32:(half + 1) * 32, r]
    return raw


def host_blob(sd, ins_num):
    flat = torch.cat([sd[k].reshape(-1) for k in weights.PARAM_KEYS]).numpy().astype(np.float64)
    idx = weights.pack_index_host(ins_num)
    return np.where(idx >= 0, flat[np.maximum(idx, 0)], 0.0), idx


def test_layout_sizes_match_library():
    lib = _lib.load()
    for ins_num in (13, 59, 93, 127, 1):
        assert lib.dmnerf_blob_floats(ins_num) == layout(ins_num)["total"]
        n_param = sum(v.numel() for v in O.make_weights(0, ins_num).values())
        assert lib.dmnerf_param_count(ins_num) == n_param
    assert lib.dmnerf_blob_floats(0) < 0 and lib.dmnerf_blob_floats(128) < 0


def test_pack_index_is_a_permutation_with_zero_padding():
    for ins_num in (13, 59, 93):
        idx = weights.pack_index_host(ins_num)
        used = idx[idx >= 0]
        n_param = _lib.load().dmnerf_param_count(ins_num)
        assert used.size == n_param and np.unique(used).size == n_param    # every parameter exactly once
        assert used.max() == n_param - 1


def test_pack_index_rejects_bad_arguments():
    lib = _lib.load()
    buf = np.empty(10, dtype=np.int32)
    assert lib.dmnerf_build_pack_index(13, buf.ctypes.data_as(ctypes.c_void_p), 10) != 0
    assert "index slots" in _lib.last_error()


def test_register_dataflow_reproduces_the_oracle_mlp():
    for ins_num, seed in ((13, 7), (59, 8)):
        sd = O.make_weights(seed, ins_num, gain=1.7)
        g = torch.Generator().manual_seed(seed)
        pts = (torch.rand(32, 3, generator=g) * 2 - 1) * 5
        dirs = torch.nn.functional.normalize(torch.randn(32, 3, generator=g), dim=-1)
        x = torch.cat([O.embed(pts, 10), O.embed(dirs, 4)], -1)
        want = O.mlp_forward({k: v.double() for k, v in sd.items()}, x.double()).numpy()
        blob, _ = host_blob(sd, ins_num)
        got = emulate_wave(blob, ins_num, x.double().numpy())
        assert got.shape == want.shape
        assert np.allclose(got, want, rtol=1e-9, atol=1e-9), np.abs(got - want).max()
