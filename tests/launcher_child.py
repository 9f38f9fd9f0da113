"""Child process of tests/test_bench_launcher.py: stands in for one rank of bench.py (gloo instead of RCCL, no GPU).
mode "ok": joins the world from the launcher's environment, all-reduces its rank and rank 0 prints one JSON line.
mode "fail": rank 1 exits with code 3 before the rendezvous, rank 0 would wait for ever (the launcher must end it)."""
import json
import os
import sys
import time

mode = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if mode == "fail":
    if rank == 1:
        sys.exit(3)
    time.sleep(600)
    sys.exit(0)
import torch
import torch.distributed as dist
dist.init_process_group("gloo", rank=rank, world_size=world)      # MASTER_ADDR / MASTER_PORT come from the launcher
t = torch.tensor([rank + 1], dtype=torch.int64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"n_gpus": world, "max": int(t.item()), "local_rank": int(os.environ["LOCAL_RANK"]),
                      "master": os.environ["MASTER_ADDR"]}), flush=True)
else:
    print("noise from a non-zero rank (must not reach the launcher's stdout)", flush=True)
dist.barrier()
dist.destroy_process_group()
