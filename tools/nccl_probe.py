"""One-rank RCCL sanity check of the calls bench.py makes when WORLD_SIZE > 1."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
h = torch.arange(256, dtype=torch.int64, device=dev)
dist.all_reduce(h, op=dist.ReduceOp.SUM)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print("ok", int(h.sum()), float(t))
dist.barrier()
dist.destroy_process_group()
