"""Can two RCCL ranks share ONE GPU on this box?  (If so, the 2-rank protocol can be exercised on a 1-GPU box.)"""
import os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    t = torch.tensor([rank + 1], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce(max) over {world} ranks on one GPU -> {int(t.item())}", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    sys.exit(3)
