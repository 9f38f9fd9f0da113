"""One rank of tests/test_gpu_fakeccl.py: the C exchange code of libsfm_amd_rccl.so (comm.cpp) with TWO ranks on ONE GPU, its RCCL
calls served by tests/fake_ccl (shared memory).  Control plane: torch.distributed over gloo.  Checks, on every rank: the sharded
estimateE (serial and pipelined) equals the single-GPU call bit for bit; sfm_process_views_sharded (views and pairs dealt
round-robin, count-sized feature exchange, record gather) equals the one-rank path; the bytes moved are what exists.
Rank 0 prints one JSON line."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import cuda_sfm_amd as S
from cuda_sfm_amd import synth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
T0 = time.time()
diag = {"rank": rank}


def stage(name, **kv):
    diag[name] = kv if kv else True
    print(f"[fakeccl_child rank {rank}/{world} +{time.time() - T0:6.2f}s] {name} {json.dumps(kv)}", file=sys.stderr, flush=True)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]


assert "fakeccl" in S.COMM_LIB_PATH, "this script is for the shared-memory stand-in only (SFM_AMD_COMM_LIB)"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)                                   # every rank on the same GPU
dist.init_process_group("gloo", rank=rank, world_size=world)
n, H = 3000, 40001
scene = synth.two_view_scene(n, seed=77)
d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
pair.fillXU(d_sift)
p = S.default_params(n, num_hypotheses=H, seed=9)
pair.estimateE(p)
ref = (pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy(), pair.get_best())
stage("single_gpu_reference", key=hex(ref[0]), best=list(ref[3]))
uid = [S.Comm.unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
comm = S.Comm(ctx, uid[0], rank, world)
ok = comm.nccl_ranks() == world
stage("communicator_up", ranks=comm.nccl_ranks())

q = S.default_params(n, num_hypotheses=H, seed=9)
q.hyp_begin, q.hyp_count = S.shard_range(H, rank, world)
key_t = torch.zeros(1, dtype=torch.int64, device=dev)
pair.ransac_score(q, key_out=key_t)
torch.cuda.synchronize()
local_key = int(key_t.item())
keys = [None] * world
dist.all_gather_object(keys, local_key)
ok = ok and max(keys) == ref[0] and len(set(keys)) == world      # two different shard keys, the larger one is the single-GPU key
stage("shard_scored", shard=[q.hyp_begin, q.hyp_count], local_key=hex(local_key))


def same_as_ref(with_key=True):
    parts = dict(key=(pair.get_key() == ref[0]) or not with_key, best=pair.get_best() == ref[3],
                 E=bool(np.array_equal(pair.get_E().view(np.uint32), ref[1].view(np.uint32))), mask=bool(np.array_equal(pair.get_inlier_mask(), ref[2])))
    if not all(parts.values()):
        stage("MISMATCH", **parts, key_now=hex(pair.get_key()), best_now=list(pair.get_best()))
    return all(parts.values())


q = S.default_params(n, num_hypotheses=H, seed=9)
comm.estimate_E(pair, q)
ok = ok and same_as_ref() and (q.hyp_begin, q.hyp_count) == S.shard_range(H, rank, world)
stage("sharded_step_done", key=hex(pair.get_key()), ok=bool(ok))
for _ in range(5):
    comm.estimate_E_pipelined(pair, q)
comm.flush()
# (the pair's own key buffer holds this rank's shard key after a pipelined step: the reduced key lives in the communicator's slot)
ok = ok and same_as_ref(with_key=False)
stage("pipelined_steps_done", ok=bool(ok))
# a second scene through the same communicator (the winner now sits in the OTHER rank's shard or not: both orders get exercised
# over the seeds), then back
for seed in (5, 6, 7):
    sc2 = synth.two_view_scene(1500, seed=seed, outlier_frac=0.5)
    p2 = S.ImagePair(ctx, sc2["K"], sc2["Kinv"], 2, 1500)
    p2.fillXU(torch.from_numpy(sc2["sift"].view(np.uint8).reshape(1500, 576)).to(dev))
    r2 = S.default_params(1500, num_hypotheses=20001, seed=seed)
    p2.estimateE(r2)
    want = (p2.get_key(), p2.get_E().copy(), p2.get_inlier_mask().copy())
    r2 = S.default_params(1500, num_hypotheses=20001, seed=seed)
    comm.estimate_E(p2, r2)
    owner = [k for k in range(world) if S.shard_range(20001, k, world)[0] <= S.unpack_key(want[0])[1] < sum(S.shard_range(20001, k, world))]
    ok = ok and p2.get_key() == want[0] and np.array_equal(p2.get_E().view(np.uint32), want[1].view(np.uint32)) and np.array_equal(p2.get_inlier_mask(), want[2])
    stage("other_scene", seed=seed, winner_owner=owner, ok=bool(ok))

# two pairs interleaved through the pipelined entry point (each step's slot, key and finalize belong to ITS pair), then a serial
# call right behind pipelined ones without an explicit flush
q = S.default_params(n, num_hypotheses=H, seed=9)
for _ in range(3):
    comm.estimate_E_pipelined(pair, q)
    comm.estimate_E_pipelined(p2, r2)
comm.flush()
good = same_as_ref(with_key=False) and np.array_equal(p2.get_E().view(np.uint32), want[1].view(np.uint32)) and np.array_equal(p2.get_inlier_mask(), want[2])
comm.estimate_E_pipelined(p2, r2)
comm.estimate_E(pair, q)                                        # (flushes the pending finalize itself)
good = good and same_as_ref() and np.array_equal(p2.get_E().view(np.uint32), want[1].view(np.uint32))
ok = ok and good
stage("interleaved_pairs", ok=bool(good))

# fewer hypotheses than ranks, and counts that do not divide: some ranks own an EMPTY shard (key 0) and still take part
for Ht in (1, 2, 3, 5, 64, 65):
    pt = S.default_params(n, num_hypotheses=Ht, seed=3)
    pair.estimateE(pt)
    want = (pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy(), pair.get_best())
    pt = S.default_params(n, num_hypotheses=Ht, seed=3)
    comm.estimate_E(pair, pt)
    good = pair.get_key() == want[0] and pair.get_best() == want[3] and np.array_equal(pair.get_E().view(np.uint32), want[1].view(np.uint32)) and np.array_equal(pair.get_inlier_mask(), want[2])
    for _ in range(3):
        comm.estimate_E_pipelined(pair, pt)
    comm.flush()
    good = good and pair.get_best() == want[3] and np.array_equal(pair.get_E().view(np.uint32), want[1].view(np.uint32)) and np.array_equal(pair.get_inlier_mask(), want[2])
    ok = ok and good
    stage("tiny_range", H=Ht, my_shard=list(S.shard_range(Ht, rank, world)), ok=bool(good))

# configs[4] inside the C library: 5 views over 2 ranks (3 + 2), one of them without a single feature, 6 pairs
w, h = 384, 288
base_d = np.array([5.0, 8.0, 12.0, 16.0, 7.0, 10.0], np.float32)
views = [synth.stereo_pair(w, h, seed=9, disparities=tuple(0.6 * k * base_d))[1] if k else synth.stereo_pair(w, h, seed=9)[0] for k in range(5)]
views.append(np.full((h, w), 100.0, np.float32))                 # a flat image: no features
K, Kinv = synth.camera(w, h)
pairs = [(0, 1), (1, 2), (4, 0), (2, 4), (3, 1), (5, 2)]
sift = dict(num_octaves=4, thresh=2.0)
vref, vcounts = S.process_views(ctx, views, K, Kinv, pairs=pairs, max_pts=4096, sift=sift, device=dev)
for rep in range(2):                                            # twice: the second call reuses (and must not trip over) the grown buffers
    vres, counts = comm.process_views(views, K, Kinv, pairs=pairs, max_pts=4096, sift=sift)
    ok = ok and counts == vcounts and sorted(vres) == sorted(vref)
    ok = ok and all(np.array_equal(vres[k].view(np.uint32), vref[k].view(np.uint32)) for k in vref)
    moved, slots_eq = comm.last_exchange()
    ok = ok and sum(counts) * 576 <= moved <= 1.1 * sum(counts) * 576 + 4096 and min(counts) == 0
    stage("views_sharded_done", rep=rep, pairs=len(vres), counts=counts, exchange_bytes=moved, slot_bytes_equivalent=slots_eq, ok=bool(ok))

# A failure on ONE rank (its slot allocation, ExtractSift, the feature buffer, sfm_process_pairs: comm.cpp's fault injection) must
# come back as an error on EVERY rank -- nobody may be left waiting in the next collective (each stage is reached by all ranks within
# the harness' time-out or the test fails) -- and the communicator must still work afterwards.
for fstage in (1, 2, 3, 4):
    os.environ["SFM_COMM_TEST_FAIL"] = f"{world - 1}:{fstage}"
    raised = False
    try:
        comm.process_views(views, K, Kinv, pairs=pairs, max_pts=4096, sift=sift)
    except S.SfmError:
        raised = True
    ok = ok and raised
    stage("injected_failure", stage=fstage, failing_rank=world - 1, this_rank_got_an_error=bool(raised))
os.environ.pop("SFM_COMM_TEST_FAIL", None)
vres, counts = comm.process_views(views, K, Kinv, pairs=pairs, max_pts=4096, sift=sift)
good = counts == vcounts and all(np.array_equal(vres[k].view(np.uint32), vref[k].view(np.uint32)) for k in vref)
ok = ok and good
stage("views_sharded_after_failures", ok=bool(good))

t = torch.tensor([1 if ok else 0], dtype=torch.int64)
dist.all_reduce(t, op=dist.ReduceOp.MIN)
everyone = [None] * world
dist.all_gather_object(everyone, diag)
if rank == 0:
    print(json.dumps({"ok": bool(t.item()), "world": world, "ranks": comm.nccl_ranks(), "per_rank": everyone}), flush=True)
comm.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
