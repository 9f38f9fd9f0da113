"""GPU: the multi-GPU estimateE protocol through the C ABI (include/sfm_amd_comm.h).  On a 1-GPU box: the 1-rank
communicator (serial and pipelined step) and the shard protocol with every 'rank' played one after the other on the
same device; with >= 2 GPUs: two real ranks over RCCL (one process per GPU, started by bench.py's launcher)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cuda_sfm_amd as S
from cuda_sfm_amd import synth
from helpers import same_bits, make_pair, to_dev
import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipelined_step_single_rank_equals_estimateE(gpu):
    """sfm_estimate_E_sharded_pipelined: scoring on the context stream, all-reduce + finalize on the exchange stream, two
    key slots.  Repeated steps (so that slots and events are re-used), then a flush: E / mask / best must equal
    sfm_estimate_E; a different seed in between must not leak into the next result."""
    torch, dev, ctx = gpu
    n, H = 3000, 50000
    scene = synth.two_view_scene(n, seed=77)
    pair, _ = make_pair(S, gpu, scene)
    refs = {}
    for seed in (9, 10):
        p = S.default_params(n, num_hypotheses=H, seed=seed)
        pair.estimateE(p)
        refs[seed] = (pair.get_best(), pair.get_E().copy(), pair.get_inlier_mask().copy())
    assert refs[9][0] != refs[10][0]
    comm = S.Comm(ctx, S.Comm.unique_id(), 0, 1)
    assert comm.nccl_ranks() == 1
    for seed in (9, 10, 10, 9, 9, 9, 10):
        q = S.default_params(n, num_hypotheses=H, seed=seed)
        comm.estimate_E_pipelined(pair, q)
    comm.flush()
    assert pair.get_best() == refs[10][0] and same_bits(pair.get_E(), refs[10][1]) and np.array_equal(pair.get_inlier_mask(), refs[10][2])
    q = S.default_params(n, num_hypotheses=H, seed=9)
    comm.estimate_E_pipelined(pair, q)
    comm.estimate_E(pair, q)                                   # the serial form flushes pending pipelined work by itself
    assert pair.get_best() == refs[9][0] and same_bits(pair.get_E(), refs[9][1])
    comm.close()


def test_process_views_sharded_single_rank(gpu):
    """sfm_process_views_sharded (configs[4] inside the C libraries: extract -> counts all-gather -> count-sized grouped
    ncclBroadcast of every view's features -> owned pairs -> ncclAllGather of records) with a 1-rank communicator against
    process_views on the same synthetic views, bit for bit; a pair list that is not the ring, a second call with fewer views
    through the cached buffers; and the exchange moves what exists (sum of count x 576), not max_pts-sized slots."""
    torch, dev, ctx = gpu
    w, h = 384, 288
    base_d = np.array([5.0, 8.0, 12.0, 16.0, 7.0, 10.0], np.float32)
    views = [synth.stereo_pair(w, h, seed=9, disparities=tuple(0.6 * k * base_d))[1] if k else synth.stereo_pair(w, h, seed=9)[0] for k in range(5)]
    K, Kinv = synth.camera(w, h)
    pairs = [(0, 1), (1, 2), (4, 0), (2, 4), (3, 1)]
    sift = dict(num_octaves=4, thresh=2.0)
    ref, rcounts = S.process_views(ctx, views, K, Kinv, pairs=pairs, max_pts=4096, sift=sift, device=dev)
    comm = S.Comm(ctx, S.Comm.unique_id(), 0, 1)
    res, counts = comm.process_views(views, K, Kinv, pairs=pairs, max_pts=4096, sift=sift)
    assert counts == rcounts and sorted(res) == sorted(ref) and len(ref) == len(pairs)
    for pid in ref:
        assert same_bits(res[pid], ref[pid]), f"pair {pid}"
    moved, slots_eq = comm.last_exchange()
    real = sum(counts) * 576
    assert real <= moved <= 1.1 * real and slots_eq == len(views) * (4096 * 576 + 64) > real
    res2, counts2 = comm.process_views(views[:3], K, Kinv, max_pts=4096, sift=sift)
    ref2, _ = S.process_views(ctx, views[:3], K, Kinv, max_pts=4096, sift=sift, device=dev)
    assert counts2 == rcounts[:3] and all(same_bits(res2[k], ref2[k]) for k in ref2) and sorted(res2) == sorted(ref2)
    more = views + [views[1], views[3]]                             # seven views: both device buffers of the communicator grow
    res3, counts3 = comm.process_views(more, K, Kinv, pairs=pairs + [(5, 6), (6, 0)], max_pts=4096, sift=sift)
    ref3, rc3 = S.process_views(ctx, more, K, Kinv, pairs=pairs + [(5, 6), (6, 0)], max_pts=4096, sift=sift, device=dev)
    assert counts3 == rc3 and sorted(res3) == sorted(ref3) and all(same_bits(res3[k], ref3[k]) for k in ref3)
    comm.close()


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("sweeps", [0, 7])
def test_estimate_E_distributed_shard_cuts_on_one_gpu(gpu, world, sweeps):
    """estimate_E_distributed with the ranks played one after the other on this GPU (score shard r -> key r; the
    'all-reduce' is max over the recorded keys; every rank finalizes from the reduced key): keys, E and mask must not
    depend on how the id range is cut."""
    torch, dev, ctx = gpu
    n, H = 2000, 20011
    scene = synth.two_view_scene(n, seed=31)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=6, jacobi_sweeps=sweeps)
    pair.estimateE(p)
    ref = (pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy(), pair.get_best())
    keys = []
    for r in range(world):                                     # pass 1: every rank's local key
        q = S.default_params(n, num_hypotheses=H, seed=6, jacobi_sweeps=sweeps)
        t = torch.zeros(1, dtype=torch.int64, device=dev)
        S.estimate_E_distributed(pair, q, r, world, t, lambda x: None)
        torch.cuda.synchronize()
        keys.append(int(t.item()))
    assert max(keys) == ref[0]
    for r in range(world):                                     # pass 2: the same step with the reduced key injected
        q = S.default_params(n, num_hypotheses=H, seed=6, jacobi_sweeps=sweeps)
        t = torch.zeros(1, dtype=torch.int64, device=dev)

        def reduce_max(x):
            x[0] = max(keys)
        S.estimate_E_distributed(pair, q, r, world, t, reduce_max)
        assert pair.get_best() == ref[3] and same_bits(pair.get_E(), ref[1]) and np.array_equal(pair.get_inlier_mask(), ref[2]), f"rank {r}"


def test_two_ranks_rccl():
    """Two processes, two GPUs, one RCCL communicator: sfm_estimate_E_sharded (+ pipelined) == the single-GPU call."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs on the node (the driver's SCALE run exercises the same path through bench.py)")
    code = ("import sys, bench; sys.exit(bench.launch_ranks(2, [], command=[sys.executable, %r], timeout=600))"
            % os.path.join(ROOT, "tests", "multi_child.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["ok"] and rec["world"] == 2 and rec["nccl_ranks"] == 2


def test_multi_rank_child_script_with_one_rank():
    """tests/multi_child.py is what runs on the multi-GPU node nobody can log into: started here with ONE rank (RCCL
    communicator of size 1) so that the script itself -- every stage, the per-rank diagnostics, the final record -- is
    exercised on every round's 1-GPU box."""
    code = ("import sys, bench; sys.exit(bench.launch_ranks(1, [], command=[sys.executable, %r], timeout=600))"
            % os.path.join(ROOT, "tests", "multi_child.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["ok"] and rec["world"] == 1 and rec["nccl_ranks"] == 1
    stages = rec["per_rank"][0]
    for name in ("process_group_up", "single_gpu_reference", "communicator_up", "shard_scored", "sharded_step_done", "pipelined_steps_done", "views_sharded_done"):
        assert name in stages, name
    assert stages["shard_scored"]["local_key"] == stages["sharded_step_done"]["key_after_allreduce"] == stages["single_gpu_reference"]["key"]
    assert "[multi_child rank 0/1" in r.stderr


def test_estimate_E_pipelined_equals_estimateE(gpu):
    """sfm_estimate_E_pipelined: consecutive calls alternate between two slots (stream + per-shard buffers) and overlap on
    the device.  Interleaved seeds and sizes, readers flush by themselves: the result is always the LAST call's, equal to
    sfm_estimate_E bit for bit; the pose stages after an explicit flush; fillXU after pipelined calls flushes too."""
    torch, dev, ctx = gpu
    n = 3000
    scene = synth.two_view_scene(n, seed=41)
    pair, d_sift = make_pair(S, gpu, scene)
    refs = {}
    for seed, H in ((9, 50000), (10, 50000), (11, 300), (12, 140000)):
        p = S.default_params(n, num_hypotheses=H, seed=seed)
        pair.estimateE(p)
        refs[(seed, H)] = (pair.get_best(), pair.get_E().copy(), pair.get_inlier_mask().copy())
    order = [(9, 50000), (10, 50000), (12, 140000), (11, 300), (10, 50000), (9, 50000), (12, 140000)]
    for upto in range(1, len(order) + 1):
        for seed, H in order[:upto]:
            pair.estimateE_pipelined(S.default_params(n, num_hypotheses=H, seed=seed))
        want = refs[order[upto - 1]]
        assert pair.get_best() == want[0] and same_bits(pair.get_E(), want[1]) and np.array_equal(pair.get_inlier_mask(), want[2]), upto
    pair.estimateE_pipelined(S.default_params(n, num_hypotheses=50000, seed=9))
    pair.flush()
    pair.computePosecandidates(); pair.choosePose(); pair.linear_triangulation()
    assert np.isfinite(pair.get_points()).all()
    for _ in range(3):
        pair.estimateE_pipelined(S.default_params(n, num_hypotheses=50000, seed=10))
    pair.fillXU(d_sift)                                       # flushes the pending steps before it rewrites the points
    pair.estimateE(S.default_params(n, num_hypotheses=50000, seed=9))
    assert pair.get_best() == refs[(9, 50000)][0]


def test_first_pipelined_call_on_the_second_slot_of_a_fresh_pair(gpu):
    """Round 6 (found by the stateful round of tests/fuzz_gpu.py): the second slot's key buffer was cleared with a plain hipMemset when it was
    allocated -- on the NULL stream, which the slot's non-blocking stream does not wait for.  With the context on the null stream (torch's default)
    the memset queued behind slot 0's kernel and cleared slot 1's key while or after its first fused-kernel launch wrote it: a partial or empty
    key reached the finalize (88 of 200 fresh pairs for two fused calls with the same solver; profiles/pipelined_burst_case.py).  Fresh pairs,
    two pipelined calls each, the kernel families that copy their key (fused) and that write it themselves (split)."""
    torch, dev, ctx = gpu
    n = 4735
    scene = synth.two_view_scene(n, seed=522527723)
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    d_sift = to_dev(torch, dev, scene["sift"])
    big, small = (20309, 3, 6.18e-7, 58246), (546, 7, 1.27e-5, 50380)
    for kernels, sweeps in (((S.KERNEL_FUSED, S.KERNEL_FUSED), (3, 7)), ((S.KERNEL_FUSED, S.KERNEL_FUSED), (0, 0)), ((S.KERNEL_SPLIT, S.KERNEL_FUSED), (3, 7)),
                            ((S.KERNEL_FUSED, S.KERNEL_SPLIT), (0, 0))):
        H, _, thr, seed = small
        key, _, oE = O.ransac_range(X0, X1, 0, H, np.float32(thr), sweeps[1], seed=seed, want_E=True)
        ocnt, ohyp = O.unpack_key(key)
        omask = O.count_inliers(oE[ohyp], X0, X1, np.float32(thr))[1]
        for rep in range(40):
            pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
            pair.fillXU(d_sift)
            for (h, _, t, sd), kern, sw in zip((big, small), kernels, sweeps):
                pair.estimateE_pipelined(S.default_params(n, num_hypotheses=h, seed=sd, kernel=kern, jacobi_sweeps=sw, threshold=t))
            assert pair.get_best() == (ohyp, ocnt), (kernels, sweeps, rep)
            assert same_bits(pair.get_E(), oE[ohyp].reshape(3, 3)) and np.array_equal(pair.get_inlier_mask(), omask)
            pair.close()
