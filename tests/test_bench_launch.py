"""``python bench.py --gpus N`` (N > 1) outside torchrun launches its own ranks (bench.self_launch).  Without a
GPU the launched ranks stop at bench.py's "needs MI355X GPUs" assertion -- which is exactly what shows that N ranks WERE launched
under torch.distributed.run with RANK / WORLD_SIZE set, and that their failure becomes the launcher's exit code with nothing on
stdout.  The GPU side of the same command: tests/test_gpu_sharded.py::test_plain_bench_command_launches_its_own_ranks."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-container form of the check")
@pytest.mark.timeout(300)
def test_plain_bench_command_spawns_n_ranks_and_relays_their_failure():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    # (the launcher stops the surviving rank with SIGTERM as soon as one fails: the message appears once or twice, the report
    # names both ranks either way)
    assert p.stderr.count("AssertionError: bench.py needs MI355X GPUs") >= 1, p.stderr[-2000:]
    assert "local_rank: 0" in p.stderr and "local_rank: 1" in p.stderr, p.stderr[-2000:]


# ---------------------------------------------------------------------------------------------------------------------------
# The contract line: round 5's bench printed one 23.7 KB JSON line and the driver recorded `parsed: null`.
# bench.contract_line builds the LAST stdout line from the full record: bounded (<= 4096 bytes), strict JSON, the task's fields.

def _strict(line):
    import json

    def no_constants(name):
        raise AssertionError(f"non-strict JSON constant {name} in the contract line")
    return json.loads(line, parse_constant=no_constants)


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "full_record")


def _r05_record():
    """Round 5's real 23.7 KB record (the line the driver could not parse), as committed under profiles/."""
    import json
    with open(os.path.join(ROOT, "profiles", "r05", "bench_r05_final.json")) as f:
        return json.load(f)


def test_contract_line_of_the_round_5_record_is_bounded_strict_and_complete():
    sys.path.insert(0, ROOT)
    import bench
    res = _r05_record()
    assert len(__import__("json").dumps(res)) > 20000              # the canned input really is the oversized one
    bench.lift_scalars(res)
    line = bench.contract_line(res, "bench_full.json")
    assert len(line.encode()) <= 4096 == bench.LINE_LIMIT and "\n" not in line
    r = _strict(line)
    for k in REQUIRED + ("cpu_baseline", "psnr_vs_oracle_db", "label_flips_vs_oracle", "speedup_vs_cpu", "train_ms_per_step", "train_rays_per_s",
                         "train_batch_rays", "train_roofline_frac_worst", "train_step_frac_of_mfma_peak"):
        assert k in r, k
    assert set(r["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"}
    assert set(r["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert "model" not in r["config"] and r["config"]["workload"]
    assert abs(r["value"] - res["value"]) <= 1e-6 * res["value"] and abs(r["roofline"]["frac"] - res["roofline"]["frac"]) <= 1e-6
    # nothing that grows: no per-kernel tables, no secondary legs, no notes
    for k in ("train", "frame", "manipulator", "train_shard_proxy", "render_split_bf16"):
        assert k not in r
    assert "all" not in r["roofline"] and "note" not in line


def test_contract_line_with_nan_and_zero_steps_is_null_not_NaN():
    sys.path.insert(0, ROOT)
    import bench
    res = _r05_record()
    res.update(value=float("nan"), ms_per_step=float("inf"), steps=0)
    res["roofline"].update(kernel_ms=float("nan"), achieved=None, frac=float("nan"))
    r = _strict(bench.contract_line(res, None))
    assert r["value"] is None and r["ms_per_step"] is None and r["roofline"]["kernel_ms"] is None and r["roofline"]["frac"] is None


def test_contract_line_at_eight_ranks_carries_the_device_table_and_stays_bounded():
    """The N > 1 line must prove what RCCL saw and still fit: 8 ranks with the longest plausible names."""
    sys.path.insert(0, ROOT)
    import bench
    res = _r05_record()
    for k in ("cpu_baseline", "psnr_vs_oracle_db", "label_flips_vs_oracle", "speedup_vs_cpu"):
        res.pop(k)
    res["n_gpus"] = 8
    res["rccl"] = {"backend": "nccl", "world_size": 8, "rccl_version": "2.27.7", "distinct_devices": 8, "one_device_dry_run": False,
                   "frame_gathers_timed": 1, "gather_send_bytes_per_rank": 2611200, "gather_bytes_per_frame": 20889600,
                   "ranks": [{"rank": r, "device_index": r, "device_name": "AMD Instinct MI355X OAM 288GB HBM3E (gfx950:sramecc+:xnack-)",
                              "pci_bus_id": "0000:%02x:00" % (5 + 16 * r), "uuid": "GPU-%032x" % (r * 0x1234567 + 99), "host": "node-with-a-long-hostname-0123",
                              "visible_devices": {"HIP_VISIBLE_DEVICES": "0,1,2,3,4,5,6,7"}, "rays_rendered": 38400} for r in range(8)]}
    res["train"].update(scaling="weak", rays_this_rank=4096, allreduce_bytes_per_step=5570704, collectives_per_step=3.0,
                        collective_kinds_per_step={"all_gather": 1.0, "all_reduce_sums": 1.0, "all_reduce_grads": 1.0},
                        collective_send_bytes_per_step=6226128.0)
    bench.lift_scalars(res)
    line = bench.contract_line(res, "bench_full.json")
    assert len(line.encode()) <= 4096
    r = _strict(line)
    assert r["rccl"]["backend"] == "nccl" and r["rccl"]["world_size"] == 8 and r["rccl"]["distinct_devices"] == 8
    assert [x["rank"] for x in r["rccl"]["ranks"]] == list(range(8))
    assert all(x["pci_bus_id"] and x["rays_rendered"] == 38400 for x in r["rccl"]["ranks"])
    assert r["train"]["allreduce_bytes_per_step"] == 5570704 and r["train"]["collectives_per_step"] == 3.0
    # 16 ranks of the same (more than one node holds) must still not break the bound
    res["rccl"]["ranks"] = [dict(res["rccl"]["ranks"][i % 8], rank=i) for i in range(16)]
    assert len(bench.contract_line(res, "bench_full.json").encode()) <= 4096
