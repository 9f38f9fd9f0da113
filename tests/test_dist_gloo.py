"""CPU, world_size 2, gloo: the multi-GPU selection protocol -- contiguous hypothesis
shards, ONE all-reduce(max) of the packed key, every rank finalizes the same winner.  The per-shard
scoring is done by the oracle here (no GPU in this container); the host logic under test is the
product's shard_range / pack_key / estimate_E_distributed control flow."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OraclePair:
    """Stands in for cuda_sfm_amd.ImagePair in estimate_E_distributed: same method names, scoring by
    the CPU oracle (test-only)."""

    def __init__(self, X0, X1):
        import oracle as O
        self.O, self.X0, self.X1 = O, X0, X1
        self.key = 0
        self.final = None

    def ransac_score(self, p, key_out=None):
        if p.hyp_count == 0:
            self.key = 0
        else:
            self.key, _, _ = self.O.ransac_range(self.X0, self.X1, p.hyp_begin, p.hyp_count, p.threshold,
                                                 p.jacobi_sweeps, seed=p.seed, want_counts=False, nthreads=2)
        if key_out is not None:                      # sfm_ransac_score_into: the key also lands in the caller's tensor
            key_out[0] = self.key

    def export_key(self, t):
        t[0] = self.key

    def ransac_finalize_key(self, p, t):
        cnt, hyp = self.O.unpack_key(int(t[0]))
        E = self.O.hypothesis_E(self.X0, self.X1, self.O.sample8(p.seed, hyp, self.X0.shape[1]), p.jacobi_sweeps)
        c, mask = self.O.count_inliers(E, self.X0, self.X1, p.threshold)
        assert c == cnt
        self.final = (hyp, cnt, E, mask)


def worker(rank, world, port, H, q):
    import sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cuda_sfm_amd as S
    import oracle as O
    from cuda_sfm_amd_synth import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.two_view_scene(300, seed=99)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    pair = OraclePair(X0, X1)
    p = S.default_params(300, num_hypotheses=H, seed=13)
    key_t = torch.zeros(1, dtype=torch.int64)
    S.estimate_E_distributed(pair, p, rank, world, key_t, lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX))
    hyp, cnt, E, mask = pair.final
    q.put((rank, hyp, cnt, E.tobytes(), mask.tobytes(), (p.hyp_begin, p.hyp_count)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [1, 97])
def test_two_rank_selection_matches_single_process(H):
    import oracle as O
    from cuda_sfm_amd_synth import synth
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, H, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sc = synth.two_view_scene(300, seed=99)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    key, counts, _ = O.ransac_range(X0, X1, 0, H, 1e-6, 7, seed=13)
    cnt, hyp = O.unpack_key(key)
    assert hyp == int(np.argmax(counts))
    assert res[0][1:5] == res[1][1:5], "ranks disagree"
    assert (res[0][1], res[0][2]) == (hyp, cnt)
    shards = [r[5] for r in res]
    assert shards[0][0] == 0 and shards[0][1] + shards[1][1] == H and shards[1][0] == shards[0][1]
