"""Does RCCL accept two ranks on ONE device?  (If it did, the product's RCCL path could be tested with 2 ranks on a 1-GPU box.)
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_same_device_probe.py"""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
    x = torch.full((4,), float(rank + 1), device="cuda:0")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce on one device -> {x.tolist()}")
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: RCCL refuses two ranks on one device: {type(e).__name__}: {str(e)[:300]}")
