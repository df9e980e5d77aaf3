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
