"""Randomised parity on whatever box runs `pytest -m gpu` (the driver's, at round end): a bounded slice of tests/fuzz_gpu.py's
rounds through the PRODUCT library (libsfm_amd.so), seeded from the committed tests/fuzz_seed.txt (bumped every round, printed
here), plus two random-seed scenes at the headline size -- 4096 matches x 2^20 hypotheses, every count -- and the reference's
ragged edge cases: num_pts2 % 32 != 0 (CudaSift/matching.cu:325, the tail FindMaxCorr10 skips) and N % 8 != 0 (SfM/kernels.h:242,
the permutation sampler's last partial slice).  Long runs: `python tests/fuzz_gpu.py <seconds> <seed> [--lib ab]`."""
import os
import time

import numpy as np
import pytest

import cuda_sfm_amd as S
from cuda_sfm_amd import synth
import oracle as O
from helpers import same_bits, to_dev, make_pair
import fuzz_gpu

pytestmark = pytest.mark.gpu
SEED = int(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_seed.txt")).read().split()[0])


def test_fuzz_slice_of_the_product_library(gpu):
    torch, dev, ctx = gpu
    assert not S.AB and S.LIB_PATH.endswith("libsfm_amd.so")
    rng = np.random.default_rng(SEED)
    table = fuzz_gpu.make_rounds(S, torch, dev, ctx, rng)
    rounds, bad, secs = fuzz_gpu.run_rounds(table, rng, budget_s=25.0, max_rounds=400, max_bad=5)
    print(f"\n[fuzz slice] seed {SEED} lib {os.path.basename(S.LIB_PATH)} rounds {rounds} in {secs:.1f} s")
    assert not bad, bad
    assert sum(rounds.values()) >= 30 and rounds["ransac"] >= 10, rounds


@pytest.mark.parametrize("k", [0, 1])
def test_random_scene_at_the_headline_size_every_count(gpu, k):
    """Not the bench's scene: noise, outlier rate, focal length, threshold and sampler seed drawn from the round's seed."""
    rng = np.random.default_rng(SEED * 7919 + k)
    n, H = 4096, 1 << 20
    sseed = int(rng.integers(1, 1 << 30))
    focal = float(rng.choice([900.0, 2360.0, 4000.0]))
    scene = synth.two_view_scene(n, seed=sseed, noise_px=float(rng.choice([0.2, 0.5, 1.5])), outlier_frac=float(rng.choice([0.1, 0.3, 0.6])), focal=focal)
    thr = float(np.float32(10.0 ** rng.uniform(-7, -5)))
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=int(rng.integers(0, 1 << 31)), threshold=thr)
    t0 = time.time()
    for _ in range(k + 1):                                       # k = 0: the first call after the fillXU (per-hypothesis operands), 1: the second (per tile)
        pair.estimateE(p)
    assert pair.last_launch()["kernel"] == S.KERNEL_PREFILTER
    assert pair.last_launch()["prefilter_rule"] == (S.PREFILTER_PER_TILE if k else S.PREFILTER_PER_HYPOTHESIS)
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    key, ocounts, _ = O.ransac_range_fast(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed)
    counts = pair.get_inlier_counts(H)
    bad = np.flatnonzero(counts != ocounts)
    print(f"\n[fuzz slice] scene seed {sseed} focal {focal} thr {thr:.3g} sampler seed {p.seed}: best {O.unpack_key(key)} in {time.time() - t0:.1f} s")
    assert bad.size == 0, f"{bad.size} of {H} counts differ, first: hyp {bad[:5]} gpu {counts[bad[:5]]} oracle {ocounts[bad[:5]]}"
    assert pair.get_key() == key
    cnt, hyp = O.unpack_key(key)
    E = O.hypothesis_E(X0, X1, O.sample8(p.seed, hyp, n), 0)
    assert same_bits(pair.get_E(), E.reshape(3, 3)) and np.array_equal(pair.get_inlier_mask(), O.count_inliers(E, X0, X1, p.threshold)[1])


def test_reference_edge_cases_ragged_sizes(gpu):
    torch, dev, ctx = gpu
    rng = np.random.default_rng(SEED + 17)
    # matcher: num_pts2 % 32 != 0 (and < 32), num_pts1 % 32 != 0
    for n1, n2 in ((int(rng.integers(33, 700)) | 1, int(rng.integers(33, 700)) | 1), (int(rng.integers(40, 200)), int(rng.integers(1, 31))), (1, 33)):
        d = synth.descriptors(max(n1, n2), seed=int(rng.integers(1, 1 << 30)), noise=0.05)
        d1 = np.ascontiguousarray(d[0][:n1]); d2 = np.ascontiguousarray(d[1][:n2])
        best = torch.empty(n1, dtype=torch.float32, device=dev); sec = torch.empty(n1, dtype=torch.float32, device=dev)
        idx = torch.empty(n1, dtype=torch.int32, device=dev)
        ctx.match_soa(to_dev(torch, dev, d1), n1, 128, to_dev(torch, dev, d2), n2, 128, best, sec, idx)
        torch.cuda.synchronize()
        ob, os_, oi = O.match_desc(d1, d2)
        assert np.array_equal(idx.cpu().numpy(), oi) and same_bits(best.cpu().numpy(), ob) and same_bits(sec.cpu().numpy(), os_), (n1, n2)
    # estimateE in reference mode: H = N / 8 disjoint slices of one permutation, N % 8 != 0
    for n in (int(rng.integers(9, 64)) | 1, int(rng.integers(2000, 2200)) * 8 + 3):
        scene = synth.two_view_scene(n, seed=int(rng.integers(1, 1 << 30)))
        pair, _ = make_pair(S, gpu, scene)
        H = n // 8
        idx = torch.empty(8 * H, dtype=torch.int32, device=dev)
        ctx.permutation_indices(n, int(rng.integers(0, 1 << 30)), idx)
        h_idx = idx.cpu().numpy().astype(np.int32).reshape(-1)
        assert h_idx.size == 8 * H and len(set(h_idx.tolist())) == 8 * H and h_idx.max() < n
        p = S.default_params(n, num_hypotheses=H, d_indices=idx)
        pair.estimateE(p)
        _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
        key, ocounts, _ = O.ransac_range(X0, X1, 0, H, p.threshold, 0, indices=h_idx)
        assert np.array_equal(pair.get_inlier_counts(H), ocounts) and pair.get_key() == key, n
