"""One rank of tests/test_gpu_multi.py::test_two_ranks_rccl (started by bench.launch_ranks, one process per GPU):
sfm_estimate_E_sharded and its pipelined form over a real RCCL communicator against the single-GPU sfm_estimate_E.
Rank 0 prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import cuda_sfm_amd as S
from cuda_sfm_amd import synth

import hashlib
import time

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
T0 = time.time()
diag = {"rank": rank}


def stage(name, **kv):
    """One line per rank and stage on stderr (flushed at once): a hang or a wrong result on a node nobody can log into is
    then diagnosable from the log alone -- which rank stopped where, and what it held at that point."""
    diag[name] = kv if kv else True
    print(f"[multi_child rank {rank}/{world} +{time.time() - T0:6.2f}s] {name} {json.dumps(kv)}", file=sys.stderr, flush=True)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]


torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
stage("process_group_up", device=torch.cuda.get_device_name(local), visible=torch.cuda.device_count())
n, H = 3000, 40001
scene = synth.two_view_scene(n, seed=77)
d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
ctx = S.Context(local, torch.cuda.current_stream().cuda_stream)
pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
pair.fillXU(d_sift)
p = S.default_params(n, num_hypotheses=H, seed=9)
pair.estimateE(p)                                               # every rank: the whole range on its own GPU
ref = (pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy(), pair.get_best())
stage("single_gpu_reference", key=hex(ref[0]), best=list(ref[3]), E=sha(ref[1]), mask=sha(ref[2]))
uid = [S.Comm.unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
stage("unique_id_received", id=hashlib.sha256(bytes(uid[0])).hexdigest()[:12])
comm = S.Comm(ctx, uid[0], rank, world)
stage("communicator_up", ncclCommCount=comm.nccl_ranks())
ok = comm.nccl_ranks() == world
# the shard's own key BEFORE the exchange (what this rank contributes to the all-reduce) ...
q = S.default_params(n, num_hypotheses=H, seed=9)
q.hyp_begin, q.hyp_count = S.shard_range(H, rank, world)
key_t = torch.zeros(1, dtype=torch.int64, device=dev)
pair.ransac_score(q, key_out=key_t)
torch.cuda.synchronize()
local_key = int(key_t.item())
stage("shard_scored", shard=[q.hyp_begin, q.hyp_count], local_key=hex(local_key), local_best=list(S.unpack_key(local_key)))
keys = [None] * world
dist.all_gather_object(keys, local_key)
ok = ok and max(keys) == ref[0]                                 # the max over the shard keys must be the single-GPU key
# ... and the exchange step itself (ncclAllReduce(max, u64) inside libsfm_amd_rccl.so + local finalize)
q = S.default_params(n, num_hypotheses=H, seed=9)
comm.estimate_E(pair, q)
stage("sharded_step_done", key_after_allreduce=hex(pair.get_key()), best=list(pair.get_best()), E=sha(pair.get_E()), mask=sha(pair.get_inlier_mask()))
ok = ok and pair.get_key() == ref[0]
ok = ok and (q.hyp_begin, q.hyp_count) == S.shard_range(H, rank, world)
ok = ok and pair.get_best() == ref[3] and np.array_equal(pair.get_E().view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(pair.get_inlier_mask(), ref[2])
for _ in range(5):                                              # pipelined: the next step's scoring overlaps this step's exchange
    comm.estimate_E_pipelined(pair, q)
comm.flush()
stage("pipelined_steps_done", best=list(pair.get_best()), E=sha(pair.get_E()), mask=sha(pair.get_inlier_mask()))
ok = ok and pair.get_best() == ref[3] and np.array_equal(pair.get_E().view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(pair.get_inlier_mask(), ref[2])
# configs[4] inside the C libraries: 5 views (spare slot on the last rank), 5 pairs, against the single-GPU path
w, h = 384, 288
base_d = np.array([5.0, 8.0, 12.0, 16.0, 7.0, 10.0], np.float32)
views = [synth.stereo_pair(w, h, seed=9, disparities=tuple(0.6 * k * base_d))[1] if k else synth.stereo_pair(w, h, seed=9)[0] for k in range(5)]
K, Kinv = synth.camera(w, h)
pairs = [(0, 1), (1, 2), (4, 0), (2, 4), (3, 1)]
sift = dict(num_octaves=4, thresh=2.0)
vref, vcounts = S.process_views(ctx, views, K, Kinv, pairs=pairs, max_pts=4096, sift=sift, device=dev)
vres, counts = comm.process_views(views, K, Kinv, pairs=pairs, max_pts=4096, sift=sift)
ok = ok and counts == vcounts and sorted(vres) == sorted(vref) and len(vref) == len(pairs)
ok = ok and all(np.array_equal(vres[k].view(np.uint32), vref[k].view(np.uint32)) for k in vref)
moved, slots_eq = comm.last_exchange()
ok = ok and sum(counts) * 576 <= moved <= 1.1 * sum(counts) * 576
stage("views_sharded_done", pairs=len(vres), counts=counts, records=sha(np.stack([vres[k] for k in sorted(vres)])) if vres else None,
      exchange_bytes=moved, slot_bytes_equivalent=slots_eq, ok=bool(ok))
# the same job through the Python harness over torch.distributed (counts all_gather, per-view broadcast, records all_gather)
def _gather(t):
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out
xs = {}
tres, tcounts = S.process_views(ctx, views, K, Kinv, pairs=pairs, rank=rank, world=world, max_pts=4096, sift=sift,
                                dist=dist if world > 1 else None, gather_results=_gather if world > 1 else None, device=dev, stats=xs)
ok = ok and tcounts == vcounts and sorted(tres) == sorted(vref) and all(np.array_equal(tres[k].view(np.uint32), vref[k].view(np.uint32)) for k in vref)
stage("views_torch_done", pairs=len(tres), exchange=xs, ok=bool(ok))
t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MIN)
diag["ok"] = bool(ok)
everyone = [None] * world
dist.all_gather_object(everyone, diag)
if rank == 0:
    import ctypes
    ctypes.CDLL(None).fflush(None)
    print(json.dumps({"ok": bool(t.item()), "world": world, "nccl_ranks": comm.nccl_ranks(), "best": list(ref[3]), "per_rank": everyone}), flush=True)
comm.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
