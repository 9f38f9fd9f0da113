"""Every inlier count of every BASELINE-size RANSAC configuration against the CPU oracle -- not a sample, not the HIP path
against itself.  oracle/sfm_oracle_fast.c is count-exact against the scalar restatement (tests/test_oracle_fast.py) and
sweeps 2^20 hypotheses x 4096 matches in about a second on the box's host cores, so the whole of

    headline  4096 matches x 2^20 hypotheses   (BASELINE metric, "4k matches"; AUTO = matrix-core pre-filter)
    C3       16384 matches x 65536 hypotheses  (BASELINE configs[2])
    C4       16384 matches x 2^20 hypotheses   (BASELINE configs[3]) -- as one call and as its eight rank shards

is compared: counts[H], the arg-max key (first maximum: highest count, lowest id -- thrust::max_element of SfM/sfm.cu:135-140
without the `-1` of :137), the winner's E bit for bit, and the inlier mask."""
import numpy as np
import pytest

import cuda_sfm_amd as S
from cuda_sfm_amd import synth
import oracle as O
from helpers import same_bits, make_pair

pytestmark = pytest.mark.gpu


def oracle_all(scene, p, H):
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    key, counts, _ = O.ransac_range_fast(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed)
    cnt, hyp = O.unpack_key(key)
    assert cnt == counts.max() and hyp == int(np.argmax(counts))          # the oracle's own key is the first maximum
    E = O.hypothesis_E(X0, X1, O.sample8(p.seed, hyp, X0.shape[1]), p.jacobi_sweeps)
    c2, mask = O.count_inliers(E, X0, X1, p.threshold)                     # the scalar restatement, for the winner
    assert c2 == cnt
    return key, counts, E, mask


def assert_all(pair, H, key, ocounts, E, mask, what):
    counts = pair.get_inlier_counts(H)
    bad = np.flatnonzero(counts != ocounts)
    assert bad.size == 0, f"{what}: {bad.size} of {H} counts differ, first: hyp {bad[:5]} gpu {counts[bad[:5]]} oracle {ocounts[bad[:5]]}"
    assert pair.get_key() == key, f"{what}: key {pair.get_key():#x} != oracle {key:#x}"
    cnt, hyp = O.unpack_key(key)
    assert pair.get_best() == (hyp, cnt)
    assert same_bits(pair.get_E(), E.reshape(3, 3)), f"{what}: E differs"
    assert np.array_equal(pair.get_inlier_mask(), mask), f"{what}: mask differs"


@pytest.mark.parametrize("name,n,H,kernel", [
    ("headline", 4096, 1 << 20, S.KERNEL_PREFILTER),
    ("c3", 16384, 65536, S.KERNEL_PREFILTER),
    ("c4", 16384, 1 << 20, S.KERNEL_PREFILTER),
])
def test_every_count_of_the_baseline_configs_equals_the_oracle(gpu, name, n, H, kernel):
    scene = synth.two_view_scene(n)                     # the scene bench.py runs
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H)           # AUTO, the bench's parameters
    pair.estimateE(p)
    assert pair.last_launch()["kernel"] == kernel, f"{name}: AUTO picked kernel {pair.last_launch()['kernel']}"
    key, ocounts, E, mask = oracle_all(scene, p, H)
    assert_all(pair, H, key, ocounts, E, mask, name)
    # the first call after the fillXU ran per-hypothesis operands (below 2^33 pairs), the second runs per-tile operands: every count again
    assert pair.last_launch()["prefilter_rule"] == (S.PREFILTER_PER_HYPOTHESIS if n * H < 2 ** 33 else S.PREFILTER_PER_TILE)
    pair.estimateE(p)
    assert pair.last_launch()["prefilter_rule"] == S.PREFILTER_PER_TILE
    assert_all(pair, H, key, ocounts, E, mask, name + " (second call)")
    # the pipelined entry point bench.py times (two slots; the finalize re-derives E from the winner's id)
    pair.estimateE_pipelined(p)
    pair.estimateE_pipelined(p)
    pair.flush()
    assert pair.get_best() == O.unpack_key(key)[::-1]                     # (d_best is what the finalize of the last step wrote)
    assert same_bits(pair.get_E(), E.reshape(3, 3)) and np.array_equal(pair.get_inlier_mask(), mask)


@pytest.mark.parametrize("kernel", [S.KERNEL_SPLIT])
def test_headline_plain_kernel_equals_the_oracle(gpu, kernel):
    """The plain wavefront kernel (what the pre-filter is compared with elsewhere) at the headline size, every count."""
    n, H = 4096, 1 << 20
    scene = synth.two_view_scene(n)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H, kernel=kernel)
    pair.estimateE(p)
    assert pair.last_launch()["kernel"] == kernel
    assert_all(pair, H, *oracle_all(scene, p, H), "headline/split")


@pytest.mark.parametrize("n,H,G", [(16384, 1 << 20, 8), (4096, 1 << 20, 8), (4096, 1 << 20, 3)])
def test_rank_shards_concatenate_to_the_oracle(gpu, n, H, G):
    """What the G ranks of a sharded run compute, one after the other on this GPU: every shard's counts against the
    oracle's slice, max of the shard keys == the oracle's key, and the finalize from the reduced key (on the LAST rank,
    whose shard does not hold the winner unless it happens to) reproduces E and mask."""
    torch, dev, ctx = gpu
    scene = synth.two_view_scene(n)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H)
    key, ocounts, E, mask = oracle_all(scene, p, H)
    key_t = torch.zeros(1, dtype=torch.int64, device=dev)
    keys = []
    covered = 0
    for r in range(G):
        b, c = S.shard_range(H, r, G)
        assert b == covered
        covered += c
        q = S.default_params(n, num_hypotheses=H, hyp_begin=b, hyp_count=c)
        pair.ransac_score(q, key_out=key_t)
        torch.cuda.synchronize()
        got = pair.get_inlier_counts(c)
        bad = np.flatnonzero(got != ocounts[b:b + c])
        assert bad.size == 0, f"shard {r}/{G}: {bad.size} counts differ, first {b + bad[:5]}"
        sk = pair.get_key()
        assert int(key_t.item()) == sk
        best = int(np.argmax(ocounts[b:b + c]))
        assert sk == O.pack_key(int(ocounts[b + best]), b + best), f"shard {r}/{G}: key is not the shard's first maximum"
        keys.append(sk)
    assert covered == H and max(keys) == key
    key_t[0] = max(keys)
    pair.ransac_finalize_key(q, key_t)
    assert pair.get_best() == O.unpack_key(key)[::-1]
    assert same_bits(pair.get_E(), E.reshape(3, 3)) and np.array_equal(pair.get_inlier_mask(), mask)


def test_finalize_refuses_caller_supplied_candidates(gpu):
    """sfm_ransac_score_candidates leaves matrices that no 8-tuple stands behind: a finalize would copy the supplied E on
    the rank that scored it and re-derive a different one elsewhere.  It must refuse (SFM_E_STATE), and the pair must
    work again after an ordinary score call."""
    torch, dev, ctx = gpu
    n, H = 600, 2048
    scene = synth.two_view_scene(n, seed=6)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=H)
    rng = np.random.default_rng(0)
    d_E = torch.from_numpy(rng.normal(size=(H, 9)).astype(np.float32)).to(dev)
    pair.ransac_score_candidates(p, d_E)
    assert pair.get_inlier_counts(H).shape == (H,)
    with pytest.raises(S.SfmError) as e:
        pair.ransac_finalize(p, 3)
    assert e.value.code == S.E_STATE
    key_t = torch.zeros(1, dtype=torch.int64, device=dev)
    pair.export_key(key_t)
    with pytest.raises(S.SfmError) as e:
        pair.ransac_finalize_key(p, key_t)
    assert e.value.code == S.E_STATE
    pair.estimateE(p)
    assert pair.get_best()[0] < H


def test_process_pairs_honours_the_match_tail_quirk_on_every_lane(gpu):
    """SFM_QUIRK_MATCH_TAIL (the reference's FindMaxCorr10 never visits the last num_pts2 % 32 points, matching.cu:325) with
    more than 8 pairs: sfm_process_pairs spreads them over four lane contexts, and every lane has to apply the quirk --
    each first view's match fields against a single quirk context's answer for the same pair."""
    torch, dev, ctx = gpu
    from helpers import to_dev
    V, n = 12, 1500 + 13                                  # n2 % 32 = 9: a tail exists
    qctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
    qctx.set_quirks(S.QUIRK_MATCH_TAIL)
    K, Kinv = synth.camera()
    views = []
    for v in range(V):
        d1, d2, perm = synth.descriptors(n, seed=40 + v)
        rec = synth.sift_records(d1 if v % 2 == 0 else d2, seed=70 + v)
        views.append(rec)
    pairs_host = [(v, (v + 1) % V) for v in range(V)]
    # the expected match fields: one quirk context, one pair at a time
    want = []
    for (i, j) in pairs_host:
        t1, t2 = to_dev(torch, dev, views[i]), to_dev(torch, dev, views[j])
        qctx.match(t1, n, t2, n)
        torch.cuda.synchronize()
        want.append(t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE).copy())
    assert any((w["match"] >= 0).all() and (w["match"] < n - n % 32).all() for w in want)
    # without the quirk some query finds its partner in the tail: the two modes really differ on this input
    plain = S.Context(0, torch.cuda.current_stream().cuda_stream)
    t1, t2 = to_dev(torch, dev, views[0]), to_dev(torch, dev, views[1])
    plain.match(t1, n, t2, n)
    torch.cuda.synchronize()
    assert not np.array_equal(t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)["match"], want[0]["match"])
    dviews = [to_dev(torch, dev, v) for v in views]
    descs = [(dviews[i], n, dviews[j], n) for (i, j) in pairs_host]
    rec, status = S.process_pairs_local(qctx, descs, K, Kinv)
    torch.cuda.synchronize()
    assert len(rec) == V
    for pid, (i, j) in enumerate(pairs_host):
        got = dviews[i].cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)
        assert np.array_equal(got["match"], want[pid]["match"]), f"pair {pid} (first view {i}): a lane ignored the quirk"
        assert same_bits(got["score"], want[pid]["score"])
    qctx.set_quirks(0)
    plain.close(); qctx.close()
