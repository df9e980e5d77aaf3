"""CPU test (no GPU: the plan is host code of the library): the split-K plans of the three weight-gradient kernels.

A plan assigns every 32-sample chunk of every (A rows x B rows) job to exactly one workgroup slice; the three cost tables
(f32 / bf16x3 / f16x2 kernels: csrc/wgrad.hip::make_plan) only change HOW MANY slices a job gets, so any plan is valid for any kernel.
Checked per (mode, ins_num, M): at most max_wgs workgroups, the slices of a job tile [0, chunks) without gap or overlap, partial
tiles do not overlap in the workspace, outputs reference their own slices, and the modes agree on everything but the slicing."""
import ctypes

import numpy as np
import pytest

from dm_nerf_amd import _lib

JOB = np.dtype([("a_off", "<i8"), ("b_off", "<i8"), ("a_R", "<i4"), ("b_R", "<i4"), ("a_row0", "<i4"), ("b_row0", "<i4"),
                ("part_off", "<i8"), ("bias_off", "<i8"), ("a_src", "<i4"), ("b_src", "<i4"), ("rowsA", "<i4"), ("rowsB", "<i4"),
                ("cls", "<i4"), ("chunk0", "<i4"), ("nchunk", "<i4"), ("pad", "<i4")])
OUT = np.dtype([("part_off", "<i8"), ("slice_stride", "<i8"), ("bias_part_off", "<i8"), ("bias_slice_stride", "<i8"),
                ("out_off", "<i8"), ("bias_out_off", "<i8"), ("n_slices", "<i4"), ("rowsA", "<i4"), ("rowsB", "<i4"), ("ldp", "<i4"),
                ("ld_out", "<i4"), ("col_off", "<i4"), ("bias_sub", "<i4"), ("ldb", "<i4"), ("perm_a", "<i4"), ("perm_b", "<i4"),
                ("to_scratch", "<i4"), ("bias_split", "<i4"), ("bias_out_off2", "<i8")])
MODES = {"f32": ("dmnerf_wgrad_plan_sizes", "dmnerf_wgrad_plan"), "bf16x3": ("dmnerf_wgrad_plan_sizes_split", "dmnerf_wgrad_plan_split"),
         "f16x2": ("dmnerf_wgrad_plan_sizes_f16", "dmnerf_wgrad_plan_f16")}


def plan(mode, ins_num, M, max_wgs):
    lib = _lib.load()
    f_sizes, f_plan = (getattr(lib, n) for n in MODES[mode])
    jb, ob, pf = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    nj, no = ctypes.c_int(), ctypes.c_int()
    _lib.check(f_sizes(ins_num, M, max_wgs, ctypes.byref(jb), ctypes.byref(ob), ctypes.byref(pf), ctypes.byref(nj), ctypes.byref(no)), "sizes")
    assert jb.value == nj.value * JOB.itemsize and ob.value == no.value * OUT.itemsize, "struct layouts of this test are stale"
    hj, ho = np.empty(jb.value, dtype=np.uint8), np.empty(ob.value, dtype=np.uint8)
    _lib.check(f_plan(ins_num, M, max_wgs, hj.ctypes.data_as(ctypes.c_void_p), jb.value, ho.ctypes.data_as(ctypes.c_void_p), ob.value), "plan")
    return hj.view(JOB), ho.view(OUT), pf.value


@pytest.mark.parametrize("ins_num,M,max_wgs", [(13, 786432, 256), (13, 262144, 256), (59, 4096 * 192, 256), (93, 3072 * 192, 256),
                                               (13, 384 * 192, 256), (1, 33, 256), (120, 5 * 50, 64), (13, 786432, 32)])
def test_plans_tile_every_job_exactly_once(ins_num, M, max_wgs):
    nchunks = ((M + 31) // 32)
    shapes = {}
    for mode in MODES:
        jobs, outs, part_floats = plan(mode, ins_num, M, max_wgs)
        assert 0 < len(jobs) <= max(max_wgs, len(outs)), (mode, len(jobs))
        assert int(outs["n_slices"].sum()) == len(jobs)
        k = 0
        spans = []
        for o in outs:
            sl = jobs[k:k + o["n_slices"]]
            k += o["n_slices"]
            # the slices of one output: same operands, consecutive chunk ranges covering [0, nchunks)
            for f in ("a_off", "b_off", "a_R", "b_R", "a_row0", "b_row0", "a_src", "b_src", "rowsA", "rowsB", "cls"):
                assert len(set(sl[f].tolist())) == 1, (mode, f)
            assert sl["chunk0"][0] == 0 and int(sl["chunk0"][-1] + sl["nchunk"][-1]) == nchunks
            assert np.array_equal(sl["chunk0"][1:], (sl["chunk0"] + sl["nchunk"])[:-1]) and bool((sl["nchunk"] >= 1).all())
            assert o["rowsA"] == sl["rowsA"][0] and o["rowsB"] == sl["rowsB"][0]
            # partial tiles: slice s at part_off + s * slice_stride, inside the workspace, not overlapping the next output's
            assert np.array_equal(sl["part_off"], o["part_off"] + o["slice_stride"] * np.arange(o["n_slices"]))
            spans.append((int(o["part_off"]), int(o["part_off"] + o["slice_stride"] * o["n_slices"])))
            has_bias = o["bias_out_off"] >= 0
            assert bool((sl["bias_off"] >= 0).all()) == bool(has_bias)
        spans.sort()
        assert spans[0][0] >= 0 and spans[-1][1] <= part_floats
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), mode
        shapes[mode] = [(int(o["out_off"]), int(o["bias_out_off"]), int(o["rowsA"]), int(o["rowsB"]), int(o["ld_out"]), int(o["col_off"]),
                         int(o["to_scratch"]), int(o["bias_split"])) for o in outs]
    assert shapes["f32"] == shapes["bf16x3"] == shapes["f16x2"]          # the same jobs; only their slicing differs


def test_the_f16_plan_gives_the_skinny_jobs_more_workgroups():
    """What the third cost table is for: with the MFMA time of the 256 x 256 jobs a fifth of the f32 kernel's, every class is priced at
    its HBM-bound chunk time and the skinny jobs (whose time did not change) get a larger share of the 256 workgroups."""
    n = {}
    for mode in MODES:
        jobs, outs, _ = plan(mode, 13, 786432, 256)
        fat = outs["n_slices"][(outs["rowsA"] == 256) & (outs["rowsB"] == 256)]
        n[mode] = (int(fat.sum()), int(outs["n_slices"].sum()))
        assert n[mode][1] == 256
    assert n["f32"][0] > n["bf16x3"][0] >= n["f16x2"][0]
