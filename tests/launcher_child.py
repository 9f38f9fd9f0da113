"""Child process of tests/test_bench_launcher.py: stands in for one rank of bench.py (gloo instead of RCCL, no GPU).
mode "ok": joins the world from the launcher's environment, all-reduces its rank and rank 0 prints one JSON line.
mode "fail": rank 1 exits with code 3 before the rendezvous, rank 0 would wait for ever (the launcher must end it).
mode "hang": every rank writes bench.py's stage lines; rank 1 then sleeps in "communicator up" (a hung ncclCommInitRank), rank 0 in
the stage after it: the launcher's time-out must report both last stages and exit 124."""
import json
import os
import sys
import time

mode = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if mode == "hang":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    bench.stage("process group up")
    bench.stage("communicator up (nccl_ranks pending)")
    if rank == 0:
        bench.stage("first step done")
    time.sleep(600)
    sys.exit(0)
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
