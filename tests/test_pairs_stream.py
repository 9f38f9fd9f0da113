"""Config 5 (many view pairs streamed over the GPUs): round-robin schedule + one gather.
CPU: schedule properties and the 2-rank gloo gather of fixed-size records; GPU: the per-pair pipeline
on a handful of synthetic pairs against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_schedule_is_a_partition():
    import cuda_sfm_amd as S
    for npairs in (0, 1, 7, 36, 630):
        for world in (1, 2, 8):
            owned = [S.pair_schedule(npairs, r, world) for r in range(world)]
            flat = sorted(x for o in owned for x in o)
            assert flat == list(range(npairs))
            assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import cuda_sfm_amd as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    npairs = 7
    max_local = (npairs + world - 1) // world
    rec = np.full((max_local, S.RESULT_FLOATS + 1), -1.0, np.float32)
    for slot, pid in enumerate(S.pair_schedule(npairs, rank, world)):
        rec[slot, :S.RESULT_FLOATS] = pid * 100 + np.arange(S.RESULT_FLOATS)
        rec[slot, S.RESULT_FLOATS] = pid
    local = torch.from_numpy(rec)
    out = torch.empty((world * max_local, S.RESULT_FLOATS + 1), dtype=torch.float32)
    dist.all_gather_into_tensor(out, local)
    got = {int(r[-1]): r[:-1].numpy().copy() for r in out if r[-1] >= 0}
    q.put((rank, sorted(got), all(np.array_equal(got[p], p * 100 + np.arange(S.RESULT_FLOATS, dtype=np.float32)) for p in got)))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_gather_of_pair_records():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60); assert p.exitcode == 0
    for rank, ids, ok in res:
        assert ids == list(range(7)) and ok


@pytest.mark.gpu
def test_pairs_pipeline_on_gpu(gpu):
    import cuda_sfm_amd as S
    import oracle as O
    from cuda_sfm_amd_synth import synth
    from helpers import same_bits, to_dev
    torch_, dev, ctx = gpu
    K, Kinv = synth.camera()
    scenes = [synth.two_view_scene(400 + 40 * i, seed=50 + i, outlier_frac=0.2) for i in range(5)]
    pairs = [(to_dev(torch_, dev, sc["sift"]), len(sc["sift"])) for sc in scenes]
    res = S.process_pairs(ctx, pairs, K, Kinv, num_hypotheses=128)
    assert sorted(res) == list(range(5))
    for i, sc in enumerate(scenes):
        _, _, X0, X1 = O.fill_xu(sc["sift"], Kinv)
        n = len(sc["sift"])
        p = S.default_params(n, num_hypotheses=128)
        key, _, Ec = O.ransac_range(X0, X1, 0, 128, p.threshold, p.jacobi_sweeps, seed=p.seed, want_E=True)
        cnt, hyp = O.unpack_key(key)
        P = O.pose_candidates(Ec[hyp], 0)
        ind, Pinv, _, _ = O.choose_pose(X0, X1, P, 0, 8)
        r = res[i]
        assert same_bits(r[:9], Ec[hyp]) and same_bits(r[9:25], Pinv[ind].reshape(16))
        assert tuple(int(v) for v in r[25:28]) == (ind, cnt, hyp)


def test_view_slots_cover_the_gathered_feature_tensor():
    """process_views: views are extracted round-robin (rank r owns r, r + world, ...) into local slots; after the
    all_gather (rank-major concatenation) every view must sit at view_slot() and nowhere else."""
    import cuda_sfm_amd as S
    for V in (1, 2, 5, 36, 37):
        for world in (1, 2, 3, 8):
            slots = (V + world - 1) // world
            gathered = np.full(world * slots, -1)
            for rank in range(world):
                for slot, v in enumerate(range(rank, V, world)):
                    gathered[rank * slots + slot] = v                     # what all_gather_into_tensor produces
            for v in range(V):
                assert gathered[S.view_slot(v, world, slots)] == v
            assert sorted(x for x in gathered if x >= 0) == list(range(V))
    assert S.ring_pairs(4) == [(0, 1), (1, 2), (2, 3), (3, 0)] and S.ring_pairs(2) == [(0, 1)] and S.ring_pairs(1) == []
    assert len(S.ring_pairs(36)) == 36


@pytest.mark.gpu
def test_process_views_ring_from_images(gpu):
    """BASELINE configs[4] from the image files on: 4 views of one scene (camera sliding along +x), ExtractSift per
    view, ring pairs (0,1) (1,2) (2,3) (3,0), MatchSiftData + the two-view pipeline per pair -- every pair's E,
    pose and support against the oracle chain run on the same images."""
    import cuda_sfm_amd as S
    import oracle as O
    from cuda_sfm_amd_synth import synth
    from helpers import same_bits
    torch, dev, ctx = gpu
    w, h = 512, 384
    base_d = np.array([5.0, 8.0, 12.0, 16.0, 7.0, 10.0], np.float32)
    views = [synth.stereo_pair(w, h, seed=9, disparities=tuple(k * base_d))[1] if k else synth.stereo_pair(w, h, seed=9)[0] for k in range(4)]
    K, Kinv = synth.camera(w, h)
    sift = dict(num_octaves=4, init_blur=1.0, thresh=2.0)
    H = 512
    res, counts = S.process_views(ctx, views, K, Kinv, max_pts=8192, sift=sift, num_hypotheses=H, pose_mode=S.POSE_CORRECT, device=dev)
    assert sorted(res) == [0, 1, 2, 3]
    feats = [O.extract_sift(v, 4, 1.0, 2.0, max_pts=8192) for v in views]
    assert counts == [f[1] for f in feats] and min(counts) > 500
    p = S.default_params(100)
    for pid, (i, j) in enumerate(S.ring_pairs(4)):
        m = O.match_sift(feats[i][0][:feats[i][1]].copy(), feats[j][0][:feats[j][1]])
        _, _, X0, X1 = O.fill_xu(m, Kinv)
        key, _, Ec = O.ransac_range(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed, want_E=True)
        ocnt, ohyp = O.unpack_key(key)
        r = res[pid]
        assert same_bits(r[:9], Ec[ohyp]) and (int(r[26]), int(r[27])) == (ocnt, ohyp)
        oP = O.pose_candidates(Ec[ohyp], S.POSE_CORRECT)
        oind, _, _, _ = O.choose_pose(X0, X1, oP, S.POSE_CORRECT, 8)
        assert int(r[25]) == oind and same_bits(r[9:25], oP[oind].reshape(16))
        if pid < 3:                                               # neighbouring views: pure +x translation, no rotation
            P = r[9:25].reshape(4, 4)
            assert np.abs(P[:3, :3] - np.eye(3)).max() < 0.03 and abs(abs(P[0, 3]) - 1.0) < 0.03


def _view_count(v):
    return 0 if v == 3 else (1 + (7 * v) % 5)                              # unequal counts, view 3 has no features at all


def _view_body(v, nbytes):
    return (np.arange(nbytes, dtype=np.int64) * (v + 1) % 251).astype(np.uint8)


def _views_worker(rank, world, port, V, max_pts, q):
    import sys
    sys.path.insert(0, ROOT)
    import cuda_sfm_amd as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    slots = (V + world - 1) // world
    rec_bytes = max_pts * 576
    block = torch.zeros((slots, rec_bytes + 64), dtype=torch.uint8)
    for slot, v in enumerate(range(rank, V, world)):                       # what sfm_extract_views leaves in the local block
        n = _view_count(v)
        block[slot, :rec_bytes] = torch.from_numpy(_view_body(v, rec_bytes))   # (bytes beyond count x 576 are stale scratch: never shipped)
        block[slot, rec_bytes:rec_bytes + 4] = torch.from_numpy(np.array([n], np.int32).view(np.uint8))

    feats, counts, offsets, stats = S.exchange_view_features(block, V, rank, world, max_pts, dist)
    ok = counts == [_view_count(v) for v in range(V)]
    real = sum(counts) * 576
    ok = ok and feats.numel() == max(real, 1) and offsets == [sum(counts[:v]) * 576 for v in range(V)]
    for v in range(V):
        nb = counts[v] * 576
        ok = ok and np.array_equal(feats[offsets[v]:offsets[v] + nb].numpy(), _view_body(v, rec_bytes)[:nb])
    # bytes on the wire: what exists (+ the counts), not max_pts-sized slots
    ok = ok and real <= stats["feature_bytes"] <= 1.1 * real and stats["slot_bytes"] == world * slots * (rec_bytes + 64) > 4 * real
    # the pair schedule on top of it: every pair owned by exactly one rank, both of its views present on that rank
    pairs = S.ring_pairs(V)
    mine = S.pair_schedule(len(pairs), rank, world)
    q.put((rank, bool(ok), mine, counts))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("V", [5, 8])
def test_two_rank_feature_exchange_of_process_views(V):
    """The feature exchange between extraction and pairing (exchange_view_features) with real slot contents, world 2, gloo:
    unequal feature counts, one view with NO features, views that do not divide evenly over the ranks.  After the exchange
    every rank holds every view's count x 576 bytes back to back in view order, the counts of all views, and what crossed the
    wire is at most 1.1 x the bytes that exist (the max_pts-sized slots of the earlier all-gather were 4.6 x on the dino ring);
    the ring pairs are dealt without overlap."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_views_worker, args=(r, 2, port, V, 16, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60); assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert sorted(res[0][2] + res[1][2]) == list(range(V)) and not set(res[0][2]) & set(res[1][2])
    assert res[0][3] == res[1][3]


def test_single_rank_exchange_is_the_block_itself():
    import cuda_sfm_amd as S
    V, max_pts = 3, 4
    block = torch.zeros((V, max_pts * 576 + 64), dtype=torch.uint8)
    for v in range(V):
        block[v, max_pts * 576:max_pts * 576 + 4] = torch.from_numpy(np.array([v + 1], np.int32).view(np.uint8))
    feats, counts, offsets, stats = S.exchange_view_features(block, V, 0, 1, max_pts)
    assert counts == [1, 2, 3] and offsets == [v * (max_pts * 576 + 64) for v in range(V)]
    assert feats.data_ptr() == block.data_ptr() and stats["feature_bytes"] == 0
