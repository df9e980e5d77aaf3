"""GPU box: the RCCL collectives the multi-GPU paths use (all_reduce f32 / f64, all_gather_into_tensor, barrier) on a
one-rank group -- checks that backend "nccl" initialises in this environment with the device_id the bench passes."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(8, dtype=torch.float32, device=dev)
dist.all_reduce(x)
s = torch.ones(4, dtype=torch.float64, device=dev)
dist.all_reduce(s)
out = torch.empty(8, 3, device=dev)
dist.all_gather_into_tensor(out, torch.ones(8, 3, device=dev))
lab = torch.empty(8, dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(lab, torch.arange(8, device=dev))
dist.barrier()
torch.cuda.synchronize()
print("rccl ok", x.tolist()[:3], float(s.sum()), float(out.sum()), int(lab.sum()))
dist.destroy_process_group()
