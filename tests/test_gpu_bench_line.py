"""The driver's N = 1 bench command on the real GPU: the LAST stdout line is the bounded contract line (<= 4096 bytes, strict JSON,
every contract field) and the full record it points at holds the rest (round 5's 23.7 KB line was recorded as
``parsed: null``).  The CPU side of the same function: tests/test_bench_launch.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strict(txt):
    def no_constants(name):
        raise AssertionError(f"non-strict JSON constant {name}")
    return json.loads(txt, parse_constant=no_constants)


@pytest.mark.timeout(900)
def test_default_bench_last_stdout_line_is_the_bounded_contract(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    full_path = os.path.join(str(tmp_path), "bench_full.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--train-steps", "2", "--full-record", full_path], capture_output=True, text=True, timeout=850, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, lines                                    # stdout = the contract line and nothing else
    assert len(lines[-1].encode()) <= 4096
    r = _strict(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "full_record", "train_ms_per_step", "train_rays_per_s", "train_batch_rays",
              "train_roofline_frac_worst", "train_step_frac_of_mfma_peak"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["unit"] == "rays/s" and r["dtype"] == "f32"
    assert r["roofline"]["bound"] == "mfma" and 0.5 < r["roofline"]["frac"] <= 1.0 and r["roofline"]["kernel_ms"] > 0
    assert abs(r["value"] - 2 * 4096 / (r["ms_per_step"] * 2e-3)) <= 1e-5 * r["value"]
    assert "rccl" not in r and "one_device_dry_run" not in r
    with open(full_path) as f:
        full = _strict(f.read())
    assert abs(full["value"] - r["value"]) <= 1e-6 * r["value"]
    for k in ("train", "train_loop", "train_shard_proxy"):           # the default legs; the opt-in ones need --extras
        assert k in full, k
    for k in ("frame", "render_split_bf16", "manipulator", "manipulator_frame", "train_split_bf16", "render_ins59"):
        assert k not in full, k
    assert len(full["train"]["roofline"]["all"]) == 3                # the per-kernel table lives in the full record
    # the stderr copy of the full record comes BEFORE the contract line would in a merged capture: it is one prefixed line
    assert any(l.startswith("bench.py full record: {") for l in p.stderr.split("\n"))
