"""CPU, world_size 2, gloo: the multi-GPU selection protocol -- contiguous hypothesis
shards, ONE all-reduce(max) of the packed key, every rank finalizes the same winner.  The per-shard
scoring is done by the oracle here (no GPU in this container); the host logic under test is the
product's shard_range / pack_key / estimate_E_distributed control flow.  (The C implementation of the same step,
sfm_estimate_E_sharded, runs on the GPU box: tests/test_gpu_ransac.py covers its shards on one device,
tests/test_gpu_multi.py its communicator with two ranks where the node has two GPUs.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 300
SCENE = dict(seed=99, outlier_frac=0.1, noise_px=0.05)          # clear inlier majority: the winner explains > half of the points


class OraclePair:
    """Stands in for cuda_sfm_amd.ImagePair in estimate_E_distributed: same method names, scoring by
    the CPU oracle (test-only).  `indices` = explicit 8-tuples in global hypothesis order (sfm_ransac_params.d_indices)."""

    def __init__(self, X0, X1, indices=None):
        import oracle as O
        self.O, self.X0, self.X1, self.indices = O, X0, X1, indices
        self.key = 0
        self.final = None

    def _tuple(self, p, hyp):
        if self.indices is not None:
            return self.indices[8 * hyp:8 * hyp + 8]
        return self.O.sample8(p.seed, hyp, self.X0.shape[1])

    def ransac_score(self, p, key_out=None):
        if p.hyp_count == 0:
            self.key = 0
        else:
            self.key, _, _ = self.O.ransac_range(self.X0, self.X1, p.hyp_begin, p.hyp_count, p.threshold,
                                                 p.jacobi_sweeps, seed=p.seed, indices=self.indices, want_counts=False, nthreads=2)
        if key_out is not None:                      # sfm_ransac_score_into: the key also lands in the caller's tensor
            key_out[0] = self.key

    def export_key(self, t):
        t[0] = self.key

    def ransac_finalize_key(self, p, t):
        cnt, hyp = self.O.unpack_key(int(t[0]))
        E = self.O.hypothesis_E(self.X0, self.X1, self._tuple(p, hyp), p.jacobi_sweeps)
        c, mask = self.O.count_inliers(E, self.X0, self.X1, p.threshold)
        assert c == cnt
        self.final = (hyp, cnt, E, mask)


def make_indices(O, X0, X1, H, sweeps, seed, case):
    """Explicit tuples that force where the winner sits.  Start from the keyed sampler's tuples, find the best one,
    then: 'rank1' -> the best tuple only at an id in rank 1's shard (its original slot gets a copy of the worst tuple);
    'tie' -> the best tuple at one id of EACH shard (equal counts across shards: the lowest id must win)."""
    idx = np.stack([O.sample8(seed, h, N) for h in range(H)]).astype(np.int32)
    _, counts, _ = O.ransac_range(X0, X1, 0, H, 1e-6, sweeps, indices=idx.reshape(-1))
    best, worst = int(np.argmax(counts)), int(np.argmin(counts))
    half = H - H // 2                                   # rank 0 owns [0, half) (shard_range gives it the extra one)
    tup = idx[best].copy()
    idx[counts == counts[best]] = idx[worst]           # no accidental co-winners
    if case == "rank1":
        want = half + (H - half) // 2
        idx[want] = tup
    else:
        want = 1
        idx[want] = tup
        idx[half + 2] = tup
    return idx.reshape(-1), want


def worker(rank, world, port, H, sweeps, case, q):
    import sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cuda_sfm_amd as S
    import oracle as O
    from cuda_sfm_amd_synth import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.two_view_scene(N, **SCENE)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    indices = None if case == "sampler" else make_indices(O, X0, X1, H, sweeps, 13, case)[0]
    pair = OraclePair(X0, X1, indices)
    p = S.default_params(N, num_hypotheses=H, seed=13, jacobi_sweeps=sweeps)
    key_t = torch.zeros(1, dtype=torch.int64)
    S.estimate_E_distributed(pair, p, rank, world, key_t, lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX))
    hyp, cnt, E, mask = pair.final
    q.put((rank, hyp, cnt, E.tobytes(), mask.tobytes(), (p.hyp_begin, p.hyp_count)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sweeps", [0, 7])                       # both null-vector solvers, the SAME setting on both sides
@pytest.mark.parametrize("H,case", [(1, "sampler"), (97, "sampler"), (64, "rank1"), (64, "tie")])
def test_two_rank_selection_matches_single_process(H, case, sweeps):
    import cuda_sfm_amd as S
    import oracle as O
    from cuda_sfm_amd_synth import synth
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, H, sweeps, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference with the SAME params object the workers build
    sc = synth.two_view_scene(N, **SCENE)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    p = S.default_params(N, num_hypotheses=H, seed=13, jacobi_sweeps=sweeps)
    indices, want = (None, None) if case == "sampler" else make_indices(O, X0, X1, H, sweeps, 13, case)
    key, counts, _ = O.ransac_range(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed, indices=indices)
    cnt, hyp = O.unpack_key(key)
    assert hyp == int(np.argmax(counts))                          # first maximum
    assert res[0][1:5] == res[1][1:5], "ranks disagree"
    assert (res[0][1], res[0][2]) == (hyp, cnt)
    shards = [r[5] for r in res]
    assert shards[0][0] == 0 and shards[0][1] + shards[1][1] == H and shards[1][0] == shards[0][1]
    if H > 1:
        assert cnt > N // 2, "the scene should have a clear winner"
    if case == "rank1":
        assert hyp == want and hyp >= shards[1][0], "winner must sit in rank 1's shard"
    if case == "tie":
        assert int((counts == cnt).sum()) == 2 and hyp == want == 1, "equal counts in both shards: the lowest id wins"
