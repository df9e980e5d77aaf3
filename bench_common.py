"""bench_common.py -- what bench.py and its leg modules (bench_train.py, bench_extras.py) share: the workload's constants (BASELINE
config 2: DM-SR 'study', 640 x 480, 64 + 128 samples, 4096-ray chunks), the MAC counts of SURVEY.md 8(d), the peaks of the MI355X
guide, and small helpers.  ``INS_NUM`` / ``MAC_PER_SAMPLE`` / ``HAVE_F16X2`` are set once by bench.main() (``--ins-num``) before any
leg runs; every module reads them as ``C.INS_NUM``."""
import json
import os
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))

INS_NUM = 13                 # DM-SR 'study' (data/color_dict.json: 13 labels)
N_RAYS = 4096                # N_test of every shipped config (configs/dmsr/train/study.txt)
N_TRAIN_SHIPPED = 3072       # N_train of the shipped train configs (configs/dmsr/train/study.txt)
S_COARSE, N_IMP = 64, 128
H_IMG, W_IMG = 480, 640
NEAR, FAR = 4.0, 15.0
MAC_PER_SAMPLE = 691712 + 128 * (INS_NUM + 1)          # 693 504
F32_MFMA_PEAK_TFLOPS = 157.3                           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
B16_MFMA_PEAK_TFLOPS = 2500.0                          # same guide: dense bf16 / f16 v_mfma_f32_32x32x16_*
HAVE_F16X2 = False                                     # set in main(): the library exports the f16x2 split kernels
HBM_PEAK_GBS, HBM_ACHIEVABLE_GBS = 8000.0, 6290.0      # same guide: HBM3E spec; the rate its own streaming benchmark measures
# HBM bytes per SAMPLE of the three training kernels at ins_num 13, from the committed rocprofv3 PMC passes of the 4096-ray step
# (profiles/r05/pmc_train_r05.txt, identical to r04's: separate --pmc runs, FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, fine launch = 786 432 samples;
# the f16x2 kernels move the same f32 rows: their counters agree within 1 %; bf16x3 saves the same rows, not separately measured).
# Counters cannot be read from inside the process, so these are NOT measured in this run: they turn a kernel time measured
# here into a GB/s figure, so that a kernel's `bound` says which roof it is actually closer to.
TRAIN_HBM_BYTES_PER_SAMPLE = {"mlp_fwd_train": (2 * 2.9427e5 + 7.6308e6) * 1e3 / 786432, "mlp_bwd_data": (2 * 3.3592e5 + 7.1332e6) * 1e3 / 786432,
                              "mlp_bwd_weights": (2 * 0.75 * 2 * 5.5508e6 + 0.75 * 2 * 63148) * 1e3 / 786432}
TRAIN_HBM_BYTES_SOURCE = "profiles/r05/pmc_train_r05.txt"
# The three kernels those byte counts were measured on.  If a kernel of that family is renamed or re-templated the counts are stale:
# train_hbm_bytes_per_sample() then returns None (no HBM view is reported) instead of pricing a new kernel with an old kernel's bytes.
TRAIN_HBM_KERNELS = ("mlp_fwd_kernel<1, false, true, false>", "mlp_bwd_kernel<1>", "wgrad_kernel")


_hbm_cache = {}


def train_hbm_bytes_per_sample():
    """TRAIN_HBM_BYTES_PER_SAMPLE, but only while the committed PMC file still names the kernels this build launches:
    every name of TRAIN_HBM_KERNELS must appear in the profile; cached."""
    if "v" not in _hbm_cache:
        ok = False
        try:
            with open(os.path.join(ROOT, TRAIN_HBM_BYTES_SOURCE)) as f:
                txt = f.read().replace(" ", "")
            ok = all(k.replace(" ", "") in txt for k in TRAIN_HBM_KERNELS)
        except OSError:
            pass
        _hbm_cache["v"] = TRAIN_HBM_BYTES_PER_SAMPLE if ok else None
    return _hbm_cache["v"]


def mac_counts(ins_num):
    """MACs per sample.  ``reference_*``: the reference's formulation (forward = wgrad = 691 712 + 128 C,
    dgrad = that - 101 248).  ``fwd`` / ``fwd_fused`` / ``dgrad`` / ``wgrad``: what this library's kernels EXECUTE after the head
    re-association (useful MACs, zero padding not counted): the fused-heads / split forward and the
    weight-gradient kernel lose the two activation-free 256 x 256 products (-131 072); the data-gradient kernel runs
    mlps.7^T .. mlps.1^T (7 x 65 536), F^T (32 768), ins_linear^T (128 C) and the two VALU heads (384 + 256)."""
    C = ins_num + 1
    ref = 691712 + 128 * C
    return {"reference_fwd": ref, "reference_dgrad": ref - 101248, "reference_wgrad": ref,
            "fwd": ref, "fwd_fused": ref - 131072, "dgrad": 7 * 65536 + 32768 + 128 * C + 384 + 256, "wgrad": ref - 131072}


def quiesce():
    """At the START of a measurement leg, before its warm-up steps: run the cyclic garbage collector now.  A full (generation-2)
    collection of a process holding a few hundred thousand Python objects pauses the host for 50-80 ms; landing inside a 20-step
    timed loop it drains the launch queue and reads as +2 .. 4 ms per step (seen on `train_loop` when an unrelated change moved the
    pause; scripts/diag_train_loop.py: the same loop is within 1 % of the resident-batch step).  A long training run pays such a
    pause once per many thousand steps.  Not placed between warm-up and timing: the GPU would sit idle for the length of the
    collection and start the timed steps from a lower clock (measured: +2 .. 5 % on the training kernels)."""
    import gc
    gc.collect()


def warm_up(one, min_steps=2, seconds=0.4):
    """Untimed warm-up of a leg: at least ``min_steps`` calls and ``seconds`` of back-to-back GPU work.  The training legs follow
    CPU baselines that leave the GPU idle for tens of seconds; the first ~100 ms of kernels after an idle period run 2-5 % slower
    (clock ramp), which two 27 ms steps do not cover."""
    t0 = time.perf_counter()
    k = 0
    while k < min_steps or time.perf_counter() - t0 < seconds:
        one()
        torch.cuda.synchronize()
        k += 1


def flush_c_stdio():
    """RCCL prints its version banner with printf (NCCL_DEBUG=VERSION on this pool); on a pipe that sits in the C
    buffer until exit and would land BEHIND the JSON line.  Push it out early instead."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                           # noqa: BLE001
        pass


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this command (counters need
    their own ``--pmc`` runs, they cannot be read from inside the process): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE,
    profiles/pmc_traffic.json -> (bytes, provenance) or (None, None)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        return d["traffic_bytes_per_launch"], d.get("source", "profiles/pmc_traffic.json")
    except Exception:                                           # noqa: BLE001
        return None, None


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:                                           # noqa: BLE001
        pass
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "torch_threads": torch.get_num_threads()}


def build_models(device, ins_num=None):
    from dm_nerf_amd import config as Cfg
    torch.manual_seed(0)
    args = types.SimpleNamespace(multires=10, multires_views=4, i_embed=0, netdepth=8, netwidth=256,
                                 ins_num=INS_NUM if ins_num is None else ins_num, device=device)
    pe, ve, mc, mf, _ = Cfg.create_nerf(args)
    with torch.no_grad():                       # "trained-like": give the density head surfaces
        mc.density_linear.bias.add_(0.3)
        mf.density_linear.bias.add_(0.3)
    return pe, ve, mc.eval(), mf.eval()


def split_products(mode):
    """16-bit MFMA products per f32 product of an ``args.mfma_split`` mode (False: the f32 MFMA, one)."""
    if not mode:
        return 1
    return 3 if str(mode) == "f16x2" else 6


def split_kernel_names(mode, obi):
    obx = 4 if obi == 3 else obi
    if str(mode) == "f16x2":
        return {"mlp_fwd_train": f"mlp_f16_kernel<{obx},true>", "mlp_bwd_data": f"mlp_bwd_f16_kernel<{obi}>",
                "mlp_bwd_weights": "wgrad_f16_kernel + reduce + unfuse"}
    return {"mlp_fwd_train": f"mlp_split_kernel<{obx},true>", "mlp_bwd_data": f"mlp_bwd_split_kernel<{obi}>",
            "mlp_bwd_weights": "wgrad_split_kernel + reduce + unfuse"}
