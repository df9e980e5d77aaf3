"""RCCL through the PRODUCT code on a one-GPU box: backend "nccl", world 1, ``device_id=`` as bench.py
passes it, and ``DMNERF_FORCE_COLLECTIVES`` semantics (dm_nerf_amd.distributed.force_collectives) so that ``all_gather_cat``
(its ``all_gather_into_tensor`` form), ``allreduce_grads(arena=)``, ``allreduce_sums`` and ``FrameRenderer.gather`` issue their
collectives instead of returning early.  The work happens in tests/_rccl_world1.py (own process); its assertions are the test,
this file checks the record it prints.  Reference: the exchanges replace networks/tester.py:63-77 (frame assembly) and the
single-process ``total_loss.backward()`` of train_dmsr.py:62-64 (gradient sum)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=600):
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.pop("DMNERF_FORCE_COLLECTIVES", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_world1.py"), *flags], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    return json.loads([l for l in p.stdout.strip().split("\n") if l.startswith("{")][-1])


@pytest.mark.timeout(900)
def test_product_collectives_on_rccl_world_one(capsys):
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    r = _run()
    with capsys.disabled():
        print(f"\n[rccl world 1] {json.dumps(r)}")
    assert r["rccl ok"] is True and r["backend"] == "nccl"
    t = r["train_step"]
    assert t["params_equal"] and t["arena_resident"] and t["arena_bytes"] == 4 * 2 * 696338
    assert r["arena_allreduce_graph"]["unchanged"] and r["arena_allreduce_graph"]["bytes"] == t["arena_bytes"]
    for k in ("band", "band_labels_only"):
        assert r[k]["equal"] and r[k]["rays"] == 38400 and r[k]["chunks"] == 10 and r[k]["gathered_is_new_buffer"]


@pytest.mark.timeout(900)
def test_whole_sharded_step_with_rccl_collectives_in_one_hip_graph(capsys):
    """GraphedTrainStep at N > 1 records the three collectives of the step inside the graph; here at world 1 on RCCL."""
    r = _run("--quick", "--graph-step")
    with capsys.disabled():
        print(f"\n[rccl world 1, graphed step] {json.dumps(r.get('graph_step'))}")
    assert r["rccl ok"] is True and len(r["graph_step"]["losses"]) == 3
