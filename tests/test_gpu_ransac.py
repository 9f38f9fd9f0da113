"""GPU parity: the HIP RANSAC path (through the C ABI) against the CPU oracle.
Integer outputs (counts, masks, winner) must be bit-exact; E is bit-exact too because both sides
follow the same arithmetic contract (tolerance stated where it is not)."""
import numpy as np
import pytest

import cuda_sfm_amd as S
from cuda_sfm_amd import synth
import oracle as O
from helpers import same_bits, to_dev, make_pair

pytestmark = pytest.mark.gpu


def oracle_xu(scene):
    return O.fill_xu(scene["sift"], scene["Kinv"])


@pytest.mark.parametrize("n", [8, 100, 128, 1000, 2048])
def test_fill_xu_bit_exact(gpu, n):
    scene = synth.two_view_scene(n, seed=11 + n)
    pair, _ = make_pair(S, gpu, scene)
    U0, U1, X0, X1 = oracle_xu(scene)
    for which, ref in ((S.BUF_U0, U0), (S.BUF_U1, U1), (S.BUF_X0, X0), (S.BUF_X1, X1)):
        assert same_bits(pair.get_XU(which), ref)


@pytest.mark.parametrize("sweeps", [0, 7])          # 0 = Householder null-vector solver, 7 = normal equations + Jacobi
@pytest.mark.parametrize("kernel", [S.KERNEL_SPLIT, S.KERNEL_FUSED])
@pytest.mark.parametrize("n,H", [(64, 50), (1000, 300), (2048, 1024), (4096, 2048), (4500, 600), (9000, 100)])
def test_counts_winner_mask_E(gpu, n, H, kernel, sweeps):
    scene = synth.two_view_scene(n, seed=5 + n)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=77, kernel=kernel, jacobi_sweeps=sweeps)
    pair.estimateE(p)
    _, _, X0, X1 = oracle_xu(scene)
    key, ocounts, oE = O.ransac_range(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=77, want_E=True)
    assert same_bits(pair.get_E_candidates(H), oE), "per-hypothesis E differs"
    assert np.array_equal(pair.get_inlier_counts(H), ocounts)
    assert pair.get_key() == key
    ocnt, ohyp = O.unpack_key(key)
    assert pair.get_best() == (ohyp, ocnt)
    assert same_bits(pair.get_E(), oE[ohyp].reshape(3, 3))
    _, omask = O.count_inliers(oE[ohyp], X0, X1, p.threshold)
    assert np.array_equal(pair.get_inlier_mask(), omask)
    assert ocnt == omask.sum()


def test_explicit_indices_and_reference_mode(gpu):
    """Reference-mode sampler (sfm.cu:95-106): H = N/8 disjoint slices of one permutation."""
    torch, dev, ctx = gpu
    n = 2048
    scene = synth.two_view_scene(n, seed=3)
    pair, _ = make_pair(S, gpu, scene)
    d_idx = torch.empty(8 * (n // 8), dtype=torch.int32, device=dev)
    ctx.permutation_indices(n, 99, d_idx)
    idx = d_idx.cpu().numpy()
    assert sorted(idx.tolist()) == list(range(n)), "not a permutation"
    p = S.default_params(n, d_indices=d_idx)
    assert p.num_hypotheses == n // 8
    pair.estimateE(p)
    _, _, X0, X1 = oracle_xu(scene)
    key, ocounts, _ = O.ransac_range(X0, X1, 0, n // 8, p.threshold, p.jacobi_sweeps, indices=idx)
    assert np.array_equal(pair.get_inlier_counts(n // 8), ocounts)
    assert pair.get_key() == key


def test_sharded_score_matches_single(gpu):
    """Hypothesis shards are independent of how they are cut."""
    n, H = 1024, 777
    scene = synth.two_view_scene(n, seed=21)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=5)
    pair.estimateE(p)
    full = pair.get_inlier_counts(H).copy()
    full_key = pair.get_key()
    keys, parts = [], []
    for r in range(3):
        b, c = S.shard_range(H, r, 3)
        q = S.default_params(n, num_hypotheses=H, seed=5, hyp_begin=b, hyp_count=c)
        pair.ransac_score(q)
        parts.append(pair.get_inlier_counts(c).copy())
        keys.append(pair.get_key())
    assert np.array_equal(np.concatenate(parts), full)
    assert max(keys) == full_key
    cnt, hyp = S.unpack_key(max(keys))
    pair.ransac_finalize(p, hyp)
    assert pair.get_best() == (hyp, cnt)


def test_degenerate_inputs(gpu):
    """All-identical correspondences (rank-deficient A) and NaN coordinates must not crash and must
    agree with the oracle (NaN residuals never count)."""
    n, H = 256, 64
    scene = synth.two_view_scene(n, seed=8)
    scene["sift"]["xpos"][:] = 100.0; scene["sift"]["ypos"][:] = 50.0
    scene["sift"]["match_xpos"][:] = 100.0; scene["sift"]["match_ypos"][:] = 50.0
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H)
    pair.estimateE(p)
    _, _, X0, X1 = oracle_xu(scene)
    key, ocounts, _ = O.ransac_range(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed)
    assert np.array_equal(pair.get_inlier_counts(H), ocounts)
    assert pair.get_key() == key


def test_errors(gpu):
    torch, dev, ctx = gpu
    scene = synth.two_view_scene(16)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, 16)
    with pytest.raises(S.SfmError) as e:
        pair.estimateE()
    assert e.value.code == S.E_STATE
    with pytest.raises(S.SfmError):
        S.ImagePair(ctx, scene["K"], scene["Kinv"], 3, 16)
    small = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, 4)
    small.fillXU(to_dev(torch, dev, scene["sift"][:4]))
    with pytest.raises(S.SfmError) as e:
        small.estimateE()
    assert e.value.code == S.E_INVALID


def test_full_size_properties(gpu):
    """BASELINE sizes (N=16384, H=65536): oracle too slow for all hypotheses -> size-independent
    properties: winner's count equals its mask sum, equals the oracle's count for that hypothesis,
    and a random sample of hypotheses matches the oracle exactly."""
    n, H = 16384, 65536
    scene = synth.two_view_scene(n)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H)
    pair.estimateE(p)
    counts = pair.get_inlier_counts(H)
    hyp, cnt = pair.get_best()
    assert cnt == counts.max() and hyp == int(np.argmax(counts))      # first maximum
    assert pair.get_inlier_mask().sum() == cnt
    _, _, X0, X1 = oracle_xu(scene)
    rng = np.random.default_rng(0)
    for h in [hyp] + rng.integers(0, H, 40).tolist():
        E = O.hypothesis_E(X0, X1, O.sample8(p.seed, h, n), p.jacobi_sweeps)
        c, _ = O.count_inliers(E, X0, X1, p.threshold, want_mask=False)
        assert c == counts[h], f"hypothesis {h}: gpu {counts[h]} oracle {c}"
    truth = ~scene["outlier"]
    m = pair.get_inlier_mask().astype(bool)
    assert (m & truth).sum() > 0.5 * truth.sum() and (m & ~truth).sum() < 0.05 * m.sum()


@pytest.mark.parametrize("n,H", [(8, 1), (9, 2), (127, 63), (129, 65), (4097, 130), (511, 8193), (70000, 40)])
@pytest.mark.parametrize("kernel", [S.KERNEL_AUTO, S.KERNEL_SPLIT, S.KERNEL_FUSED])
def test_ragged_sizes_all_kernels(gpu, n, H, kernel):
    """Ragged / extreme sizes: every kernel family agrees with the oracle (counts, winner, mask, E)."""
    scene = synth.two_view_scene(n, seed=1000 + n)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=n + H, kernel=kernel)
    pair.estimateE(p)
    _, _, X0, X1 = oracle_xu(scene)
    if n <= 5000:
        key, ocounts, oE = O.ransac_range(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=n + H, want_E=True)
        assert np.array_equal(pair.get_inlier_counts(H), ocounts)
        assert pair.get_key() == key
        ocnt, ohyp = O.unpack_key(key)
        assert same_bits(pair.get_E(), oE[ohyp].reshape(3, 3))
        assert np.array_equal(pair.get_inlier_mask(), O.count_inliers(oE[ohyp], X0, X1, p.threshold)[1])
    else:
        counts = pair.get_inlier_counts(H)
        for h in range(0, H, 7):
            E = O.hypothesis_E(X0, X1, O.sample8(n + H, h, n), p.jacobi_sweeps)
            assert O.count_inliers(E, X0, X1, p.threshold, want_mask=False)[0] == counts[h]
        hyp, cnt = pair.get_best()
        assert cnt == counts.max() and hyp == int(np.argmax(counts)) and pair.get_inlier_mask().sum() == cnt


def test_set_points_generic_z(gpu):
    """sfm_set_points with arbitrary homogeneous scale (z != 1) takes the generic scoring kernel."""
    torch, dev, ctx = gpu
    n, H = 1500, 9000
    scene = synth.two_view_scene(n, seed=4)
    _, _, X0, X1 = oracle_xu(scene)
    w0 = (1.0 + 0.5 * synth.uniform01(7, n)).astype(np.float32); w1 = (2.0 - synth.uniform01(8, n)).astype(np.float32)
    X0s = np.ascontiguousarray(X0 * w0, np.float32); X1s = np.ascontiguousarray(X1 * w1, np.float32)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.set_points(to_dev(torch, dev, X0s), to_dev(torch, dev, X1s))
    for kernel in (S.KERNEL_SPLIT, S.KERNEL_FUSED):
        p = S.default_params(n, num_hypotheses=H, seed=3, kernel=kernel)
        pair.estimateE(p)
        key, ocounts, _ = O.ransac_range(X0s, X1s, 0, H, p.threshold, p.jacobi_sweeps, seed=3)
        assert np.array_equal(pair.get_inlier_counts(H), ocounts) and pair.get_key() == key


@pytest.mark.parametrize("scale", [3.0e4, 2.0e5, 1.0e9])
def test_huge_coordinates_keep_the_full_range_tracking(gpu, scale):
    """The scoring kernel drops the upper range check of thr*da*db only when every |coordinate| < 1e5 (then it
    cannot overflow).  Points in pixel-like or absurd units (homogeneous scale is free) must still count exactly
    like the oracle, through the variant that keeps the check and, for 1e9, through the exact fallback."""
    torch, dev, ctx = gpu
    n, H = 1100, 3000
    scene = synth.two_view_scene(n, seed=14)
    _, _, X0, X1 = oracle_xu(scene)
    X0s = np.ascontiguousarray(X0 * np.float32(scale), np.float32); X1s = np.ascontiguousarray(X1 * np.float32(scale), np.float32)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.set_points(to_dev(torch, dev, X0s), to_dev(torch, dev, X1s))
    thr = float(np.float32(1e-6) * np.float32(scale) * np.float32(scale)) if scale < 1e8 else 1e-6
    for kernel in (S.KERNEL_SPLIT, S.KERNEL_FUSED):
        p = S.default_params(n, num_hypotheses=H, seed=5, kernel=kernel, threshold=thr)
        pair.estimateE(p)
        key, ocounts, _ = O.ransac_range(X0s, X1s, 0, H, p.threshold, p.jacobi_sweeps, seed=5)
        assert np.array_equal(pair.get_inlier_counts(H), ocounts) and pair.get_key() == key


def test_rccl_comm_single_rank_equals_estimateE(gpu):
    """include/sfm_amd_comm.h with one rank: shard = everything, the all-reduce is the identity; results must equal
    sfm_estimate_E (the N > 1 behaviour is the same code with smaller shards; covered by tests/test_dist_gloo.py on the
    host logic and measured by bench.py --gpus N --comm rccl)."""
    torch, dev, ctx = gpu
    n, H = 3000, 5000
    scene = synth.two_view_scene(n, seed=77)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=9)
    pair.estimateE(p)
    ref = (pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy(), pair.get_inlier_counts(H).copy())
    comm = S.Comm(ctx, S.Comm.unique_id(), 0, 1)
    q = S.default_params(n, num_hypotheses=H, seed=9)
    comm.estimate_E(pair, q)
    assert (q.hyp_begin, q.hyp_count) == (0, H)
    assert pair.get_key() == ref[0] and same_bits(pair.get_E(), ref[1]) and np.array_equal(pair.get_inlier_mask(), ref[2])
    assert np.array_equal(pair.get_inlier_counts(H), ref[3])
    comm.close()


@pytest.mark.parametrize("n,H", [(1000, 300000), (5000, 200000)])
def test_many_hypotheses_oversubscribed_grid(gpu, n, H):
    """Above 64k hypotheses the scoring grid is larger than what is co-resident (16 blocks per CU, each block running
    several batches over its staged tile; with n > 4096 also over several tiles): every count against the oracle."""
    scene = synth.two_view_scene(n, seed=300 + n)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=12, kernel=S.KERNEL_SPLIT)
    pair.estimateE(p)
    assert pair.last_launch()["grid"] > 512
    _, _, X0, X1 = oracle_xu(scene)
    key, ocounts, _ = O.ransac_range(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=12)
    assert np.array_equal(pair.get_inlier_counts(H), ocounts)
    assert pair.get_key() == key


@pytest.mark.parametrize("kernel,H", [(S.KERNEL_SPLIT, 40000), (S.KERNEL_SPLIT, 3000), (S.KERNEL_FUSED, 500)])
def test_score_into_leaves_the_key_in_caller_memory(gpu, kernel, H):
    """sfm_ransac_score_into == sfm_ransac_score + sfm_ransac_export_key, for every kernel family, also on a buffer that
    holds garbage before the call and across repeated calls (the keys are cleared inside the call)."""
    torch, dev, ctx = gpu
    n = 1200
    scene = synth.two_view_scene(n, seed=71)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, seed=4, kernel=kernel)
    pair.ransac_score(p)
    ref = torch.zeros(1, dtype=torch.int64, device=dev)
    pair.export_key(ref)
    torch.cuda.synchronize()
    for _ in range(3):
        out = torch.full((1,), 0x7FFFFFFFFFFFFFFF, dtype=torch.int64, device=dev)
        pair.ransac_score(p, key_out=out)
        pair.ransac_finalize_key(p, out)
        torch.cuda.synchronize()
        assert int(out.item()) == int(ref.item()) == pair.get_key()
        assert pair.get_best() == tuple(reversed(S.unpack_key(int(ref.item()))))
    q = S.default_params(n, num_hypotheses=H, seed=4, kernel=kernel)
    q.hyp_begin, q.hyp_count = H, 0                       # empty shard: key 0
    out = torch.full((1,), 5, dtype=torch.int64, device=dev)
    pair.ransac_score(q, key_out=out)
    torch.cuda.synchronize()
    assert int(out.item()) == 0


def test_generic_z_two_hypotheses_per_wavefront(gpu):
    """Homogeneous coordinates with z != 1 and enough hypotheses for the two-per-wavefront scoring kernel (generic 96 KiB
    tile layout): every count against the oracle."""
    torch, dev, ctx = gpu
    n, H = 1300, 50001                                     # odd count: the last wavefront scores a single hypothesis
    scene = synth.two_view_scene(n, seed=44)
    _, _, X0, X1 = oracle_xu(scene)
    X0s = np.ascontiguousarray(X0 * (0.5 + synth.uniform01(3, n)).astype(np.float32))
    X1s = np.ascontiguousarray(X1 * (2.0 - synth.uniform01(4, n)).astype(np.float32))
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.set_points(to_dev(torch, dev, X0s), to_dev(torch, dev, X1s))
    p = S.default_params(n, num_hypotheses=H, seed=8)
    pair.estimateE(p)
    key, ocounts, _ = O.ransac_range(X0s, X1s, 0, H, p.threshold, p.jacobi_sweeps, seed=8)
    assert np.array_equal(pair.get_inlier_counts(H), ocounts) and pair.get_key() == key


def _oracle_sample_check(X0, X1, p, counts, hyps, n, base=0):
    for h in hyps:
        E = O.hypothesis_E(X0, X1, O.sample8(p.seed, int(h), n), p.jacobi_sweeps)
        c, _ = O.count_inliers(E, X0, X1, p.threshold, want_mask=False)
        assert c == counts[int(h) - base], f"hypothesis {h}: gpu {counts[int(h) - base]} oracle {c}"


def test_c4_full_size_single_gpu(gpu):
    """BASELINE configs[3] on ONE GPU: 16384 matches x 2^20 hypotheses.  The oracle needs ~1.5 ms per hypothesis at this
    size, so: sampled counts against the oracle, winner = first arg-max of ALL counts, mask sum = count, and both
    scoring arrangements (matrix-core pre-filter = AUTO; plain wavefront kernel with the tile-parallel grid; the in-block tile
    loop of round 1 is an A/B variant, tests/test_gpu_ab.py) give the same 2^20 counts and the same key."""
    n, H = 16384, 1 << 20
    scene = synth.two_view_scene(n)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H)
    pair.estimateE(p)
    assert pair.last_launch()["kernel"] == S.KERNEL_PREFILTER          # what AUTO picks at this size (matrix-core pre-filter)
    counts = pair.get_inlier_counts(H).copy()
    key = pair.get_key()
    hyp, cnt = pair.get_best()
    assert cnt == counts.max() and hyp == int(np.argmax(counts))
    assert key == S.pack_key(cnt, hyp)
    mask = pair.get_inlier_mask()
    assert mask.sum() == cnt
    _, _, X0, X1 = oracle_xu(scene)
    rng = np.random.default_rng(4)
    top = np.argsort(counts)[-8:]                                 # the best few and a random sample
    _oracle_sample_check(X0, X1, p, counts, [hyp] + top.tolist() + rng.integers(0, H, 48).tolist(), n)
    E = O.hypothesis_E(X0, X1, O.sample8(p.seed, hyp, n), p.jacobi_sweeps)
    assert same_bits(pair.get_E(), E.reshape(3, 3))
    assert np.array_equal(mask, O.count_inliers(E, X0, X1, p.threshold)[1])
    q = S.default_params(n, num_hypotheses=H, kernel=S.KERNEL_SPLIT)   # the plain wavefront kernel (tile-parallel grid)
    pair.estimateE(q)
    assert pair.last_launch()["kernel"] == S.KERNEL_SPLIT
    assert np.array_equal(pair.get_inlier_counts(H), counts) and pair.get_key() == key


def test_c4_eight_shards_equal_the_single_call(gpu):
    """BASELINE configs[3] as the 8 ranks see it: shards hyp_begin = r * 2^17 of the same 16384-match scene, scored one
    after the other on this GPU.  Concatenated counts == the single call's counts, max of the shard keys == its key,
    and finalizing from the reduced key reproduces E / mask / best bit for bit."""
    torch, dev, ctx = gpu
    n, H, G = 16384, 1 << 20, 8
    scene = synth.two_view_scene(n)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H)
    pair.estimateE(p)
    full = pair.get_inlier_counts(H).copy()
    ref = (pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy(), pair.get_best())
    keys = []
    key_t = torch.zeros(1, dtype=torch.int64, device=dev)
    for r in range(G):
        b, c = S.shard_range(H, r, G)
        assert (b, c) == (r << 17, 1 << 17)
        q = S.default_params(n, num_hypotheses=H, hyp_begin=b, hyp_count=c)
        pair.ransac_score(q, key_out=key_t)
        torch.cuda.synchronize()
        assert np.array_equal(pair.get_inlier_counts(c), full[b:b + c]), f"shard {r}"
        assert int(key_t.item()) == pair.get_key()
        keys.append(pair.get_key())
    assert max(keys) == ref[0]
    key_t[0] = max(keys)                                     # what the all-reduce(max) leaves on every rank
    pair.ransac_finalize_key(q, key_t)                       # the last rank's shard does not hold the winner's E: re-derived
    assert pair.get_best() == ref[3] and same_bits(pair.get_E(), ref[1]) and np.array_equal(pair.get_inlier_mask(), ref[2])


def test_finalize_from_a_key_that_names_no_hypothesis(gpu):
    """A key of 0 (every shard empty / uninitialised buffer / failed all-reduce) decodes to hypothesis 0xFFFFFFFF: the
    finalize step must not index the tuple table with it; result is defined (E = 0, empty mask) and reported."""
    torch, dev, ctx = gpu
    n = 500
    scene = synth.two_view_scene(n, seed=2)
    pair, _ = make_pair(S, gpu, scene)
    d_idx = torch.zeros(8 * 10, dtype=torch.int32, device=dev)
    d_idx[:] = torch.arange(80, dtype=torch.int32, device=dev) % n
    p = S.default_params(n, num_hypotheses=10, d_indices=d_idx)
    key_t = torch.zeros(1, dtype=torch.int64, device=dev)
    pair.ransac_finalize_key(p, key_t)
    with pytest.raises(S.SfmError) as e:
        pair.get_best()
    assert e.value.code == S.E_STATE
    assert pair.get_inlier_mask().sum() == 0 and not pair.get_E().any()
    key_t[0] = S.pack_key(3, 10)                              # id == num_hypotheses: out of range too
    pair.ransac_finalize_key(p, key_t)
    with pytest.raises(S.SfmError):
        pair.get_best()
    pair.estimateE(p)                                          # and the pair is still usable
    assert pair.get_best()[0] < 10



def test_a_context_outlives_its_destroy_call_while_pairs_point_at_it(gpu):
    """sfm_ctx_destroy on a context with live pairs only marks it (include/sfm_amd.h): a host language whose finalizers run in no
    particular order -- Python's cyclic collector at interpreter exit did, and an Image_pair then synchronised a freed context's stream
    -- cannot leave a pair with a dangling pointer.  The pair keeps working; the last pair to go takes the context down."""
    torch, dev, _ = gpu
    n = 600
    scene = synth.two_view_scene(n, seed=31)
    ctx2 = S.Context(0, torch.cuda.current_stream().cuda_stream)
    pair_a, d_sift = make_pair(S, (torch, dev, ctx2), scene)
    pair_b = S.ImagePair(ctx2, scene["K"], scene["Kinv"], 2, n)
    ctx2.close()                                            # the owner lets go first
    assert ctx2._h is None
    p = S.default_params(n, num_hypotheses=300, seed=4)
    pair_a.estimateE(p)                                     # ... and the pair still has a valid context under it
    _, _, X0, X1 = oracle_xu(scene)
    key, ocounts, _ = O.ransac_range(X0, X1, 0, 300, p.threshold, 0, seed=4)
    assert pair_a.get_key() == key and np.array_equal(pair_a.get_inlier_counts(300), ocounts)
    pair_b.fillXU(d_sift)
    pair_a.close()
    pair_b.estimateE(p)
    assert pair_b.get_key() == key
    pair_b.close()                                          # the last reference: the context goes with it
