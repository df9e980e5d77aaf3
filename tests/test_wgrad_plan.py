"""CPU test (no GPU: the plan is host code of the library): the split-K plans of the three weight-gradient kernels.

A plan assigns every 32-sample chunk of every (A rows x B rows) job to exactly one work item (a sample slice with its own
partial tile) and every item to a workgroup: a LEADER item (``follow`` >= 0) and the ``follow`` items behind it run in one
workgroup, filled to the same time as every other (csrc/wgrad.hip::make_plan).  The three cost tables (f32 / bf16x3 / f16x2
kernels) only change WHERE the slices are cut, so any plan is valid for any kernel.
Checked per (mode, ins_num, M): at most max_wgs workgroups, leader / follower chains consistent, the slices of a job tile
[0, chunks) without gap or overlap, partial tiles do not overlap in the workspace, outputs reference their own slices, the
workgroups' modelled times are balanced, and the modes agree on everything but the slicing."""
import ctypes

import numpy as np
import pytest

from dm_nerf_amd import _lib

JOB = np.dtype([("a_off", "<i8"), ("b_off", "<i8"), ("a_R", "<i4"), ("b_R", "<i4"), ("a_row0", "<i4"), ("b_row0", "<i4"),
                ("part_off", "<i8"), ("bias_off", "<i8"), ("a_src", "<i4"), ("b_src", "<i4"), ("rowsA", "<i4"), ("rowsB", "<i4"),
                ("cls", "<i4"), ("chunk0", "<i4"), ("nchunk", "<i4"), ("follow", "<i4"), ("next", "<i4"), ("pad", "<i4")])
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
    assert jb.value % JOB.itemsize == 0 and jb.value >= nj.value * JOB.itemsize and ob.value == no.value * OUT.itemsize, "struct layouts of this test are stale"
    hj, ho = np.empty(jb.value, dtype=np.uint8), np.empty(ob.value, dtype=np.uint8)
    _lib.check(f_plan(ins_num, M, max_wgs, hj.ctypes.data_as(ctypes.c_void_p), jb.value, ho.ctypes.data_as(ctypes.c_void_p), ob.value), "plan")
    jobs = hj.view(JOB)
    # table = [one leader per workgroup (the grid: n_jobs blocks)] + [follower items]; a leader's followers are consecutive from `next`
    n_wgs = nj.value
    assert bool((jobs["follow"][:n_wgs] >= 0).all()) and bool((jobs["follow"][n_wgs:] == -1).all())
    nxt = n_wgs
    for i in range(n_wgs):
        if jobs["follow"][i] > 0:
            assert jobs["next"][i] == nxt
            nxt += int(jobs["follow"][i])
        else:
            assert jobs["next"][i] == -1
    assert nxt == len(jobs)
    return jobs, ho.view(OUT), pf.value, n_wgs


def by_output(jobs, outs):
    """The items of every output in slice order (they are addressed through their partial tiles, not their table position)."""
    groups = []
    for o in outs:
        sel = (jobs["part_off"] >= o["part_off"]) & (jobs["part_off"] < o["part_off"] + o["slice_stride"] * o["n_slices"])
        sl = jobs[sel]
        groups.append(sl[np.argsort(sl["part_off"])])
    return groups


def wg_items(jobs, n_wgs):
    """Per workgroup: its items in the order it runs them."""
    return [np.concatenate([jobs[i:i + 1], jobs[jobs["next"][i]:jobs["next"][i] + jobs["follow"][i]]]) if jobs["follow"][i] > 0 else jobs[i:i + 1]
            for i in range(n_wgs)]


@pytest.mark.parametrize("ins_num,M,max_wgs", [(13, 786432, 256), (13, 262144, 256), (59, 4096 * 192, 256), (93, 3072 * 192, 256),
                                               (13, 384 * 192, 256), (1, 33, 256), (120, 5 * 50, 64), (13, 786432, 32)])
def test_plans_tile_every_job_exactly_once(ins_num, M, max_wgs):
    nchunks = ((M + 31) // 32)
    shapes = {}
    for mode in MODES:
        jobs, outs, part_floats, n_wgs = plan(mode, ins_num, M, max_wgs)
        assert 0 < n_wgs <= max_wgs, (mode, n_wgs)
        assert int(outs["n_slices"].sum()) == len(jobs)
        spans = []
        for o, sl in zip(outs, by_output(jobs, outs)):
            assert len(sl) == o["n_slices"]
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


CHUNK_NS = {"f32": (7307, 3680, 1950, 600, 1060, 600, 1100, 1500, 1900),          # csrc/wgrad.hip::make_plan: measured ns per 32-sample
            "bf16x3": (4600, 2700, 2030, 895, 1330, 975, 1180, 1390, 1710),       # chunk of each shape class, per kernel
            "f16x2": (2811, 2100, 1341, 660, 950, 738, 1043, 1217, 1391)}


def _wg_times(mode, jobs, n_wgs):
    ns = np.array(CHUNK_NS[mode], dtype=np.float64)
    return np.array([float((ns[it["cls"]] * it["nchunk"]).sum()) for it in wg_items(jobs, n_wgs)])


@pytest.mark.parametrize("M", [786432, 3072 * 192, 384 * 192])
def test_workgroups_are_filled_to_the_same_time(M):
    """What the wrap-around packing is for: all 256 workgroups busy, the longest within 1 % of the mean of the modelled chunk times
    (the one-slice-per-workgroup plan of earlier rounds: 3 % at 786 432 samples), at most three items per workgroup."""
    for mode in MODES:
        jobs, outs, _, n_wgs = plan(mode, 13, M, 256)
        t = _wg_times(mode, jobs, n_wgs)
        assert 250 <= len(t) <= 256
        # (chunk times only: the plan also charges ~10 us per item for the ring fill and the tile store, 2 - 4 % of a 0.26 - 0.57 ms launch)
        assert t.max() <= (1.01 if M >= 500000 else 1.05) * t.mean(), (mode, t.max() / t.mean())
        assert int(jobs["follow"].max()) <= 2


def test_the_f16_plan_gives_the_skinny_jobs_more_of_the_launch():
    """What the third cost table is for: with the MFMA time of the 256 x 256 jobs a fifth of the f32 kernel's, every class is priced at
    its HBM-bound chunk time and the skinny jobs (whose time did not change) get a larger share of the workgroups' time: the fat
    jobs are cut into fewer slices."""
    n = {}
    for mode in MODES:
        jobs, outs, _, _ = plan(mode, 13, 786432, 256)
        fat = outs["n_slices"][(outs["rowsA"] == 256) & (outs["rowsB"] == 256)]
        n[mode] = int(fat.sum())
    assert n["f32"] > n["bf16x3"] >= n["f16x2"]
