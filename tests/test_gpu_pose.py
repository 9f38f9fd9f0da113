"""GPU parity: pose candidates, choosePose and linear triangulation against the oracle."""
import numpy as np
import pytest

import cuda_sfm_amd as S
from cuda_sfm_amd import synth
import oracle as O
from helpers import same_bits, make_pair, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [S.POSE_REFERENCE, S.POSE_CORRECT])
@pytest.mark.parametrize("n", [200, 2048])
def test_pose_pipeline(gpu, n, mode):
    scene = synth.two_view_scene(n, seed=40 + n, outlier_frac=0.2)
    pair, _ = make_pair(S, gpu, scene)
    p = S.default_params(n, num_hypotheses=512)
    pair.estimateE(p)
    pair.computePosecandidates(mode)
    pair.choosePose(mode)
    pair.linear_triangulation(mode)

    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    E = pair.get_E()
    oP = O.pose_candidates(E, mode)
    assert same_bits(pair.get_pose_candidates(), oP)
    oind, oPinv, _, _ = O.choose_pose(X0, X1, oP, mode, sweeps=8)
    assert pair.get_pose_index() == oind
    assert same_bits(pair.get_pose_inverses(), oPinv)
    Pm = oPinv[oind] if mode == S.POSE_REFERENCE else oP[oind]
    opts = O.triangulate(X0, X1, Pm, sweeps=8)
    assert same_bits(pair.get_points(), opts)


@pytest.mark.parametrize("mode", [S.POSE_REFERENCE, S.POSE_CORRECT])
@pytest.mark.parametrize("n", [8, 47, 48, 49, 200, 2048, 5000])
def test_pose_chain_equals_three_calls(gpu, n, mode):
    """sfm_pose_chain (one launch in REFERENCE mode) against computePosecandidates + choosePose + linear_triangulation on
    the same pair and against the oracle: candidates, inverses, index, votes and points bit for bit; get_result too."""
    scene = synth.two_view_scene(n, seed=140 + n, outlier_frac=0.2)
    pair, _ = make_pair(S, gpu, scene)
    pair.estimateE(S.default_params(n, num_hypotheses=256))
    pair.computePosecandidates(mode); pair.choosePose(mode); pair.linear_triangulation(mode)
    want = (pair.get_pose_candidates(), pair.get_pose_inverses(), pair.get_pose_index(), pair.get_points(), pair.get_result())
    pair.estimateE(S.default_params(n, num_hypotheses=256))           # clears the pose state
    with pytest.raises(S.SfmError):
        pair.get_points()
    pair.pose_chain(mode)
    got = (pair.get_pose_candidates(), pair.get_pose_inverses(), pair.get_pose_index(), pair.get_points(), pair.get_result())
    assert got[2] == want[2]
    for g, w in zip(got, want):
        assert same_bits(np.asarray(g, np.float32), np.asarray(w, np.float32))
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    oP = O.pose_candidates(pair.get_E(), mode)
    oind, oPinv, _, _ = O.choose_pose(X0, X1, oP, mode, sweeps=8)
    assert got[2] == oind and same_bits(got[0], oP) and same_bits(got[1], oPinv)
    assert same_bits(got[3], O.triangulate(X0, X1, oPinv[oind] if mode == S.POSE_REFERENCE else oP[oind], sweeps=8))


def test_pose_chain_needs_E(gpu):
    pair, _ = make_pair(S, gpu, synth.two_view_scene(64))
    with pytest.raises(S.SfmError) as e:
        pair.pose_chain()
    assert e.value.code == S.E_STATE


def test_correct_mode_recovers_ground_truth(gpu):
    """Noise-free scene: CORRECT mode must give the true R, t (up to scale) and exact depths.
    Tolerances: rotation 5e-3, translation direction 5e-3, points 1e-2 relative (fp32 DLT)."""
    n = 1024
    scene = synth.two_view_scene(n, seed=9, noise_px=0.0, outlier_frac=0.0)
    pair, _ = make_pair(S, gpu, scene)
    pair.estimateE(S.default_params(n, num_hypotheses=256))
    assert pair.get_best()[1] > 0.95 * n
    pair.computePosecandidates(S.POSE_CORRECT); pair.choosePose(S.POSE_CORRECT); pair.linear_triangulation(S.POSE_CORRECT)
    P = pair.get_pose_candidates()[pair.get_pose_index()].astype(np.float64)
    R, t = P[:3, :3], P[:3, 3]
    assert np.abs(R - scene["R"]).max() < 5e-3
    assert np.abs(t / np.linalg.norm(t) - scene["t"]).max() < 5e-3
    pts = pair.get_points()[:3].T.astype(np.float64)
    gt = scene["points3d"]
    scale = np.median(np.linalg.norm(gt, axis=1) / np.linalg.norm(pts, axis=1))
    assert np.median(np.linalg.norm(pts * scale - gt, axis=1) / np.linalg.norm(gt, axis=1)) < 1e-2


def test_call_order_errors(gpu):
    torch, dev, ctx = gpu
    scene = synth.two_view_scene(64)
    pair, _ = make_pair(S, gpu, scene)
    for fn in (pair.computePosecandidates, pair.choosePose, pair.linear_triangulation):
        with pytest.raises(S.SfmError) as e:
            fn()
        assert e.value.code == S.E_STATE


def test_copy_points_to_vbo(gpu):
    """Image_pair::copyBoidsToVBO (sfm.cu:374-383): interleaved (x, y, z, 1) * scale and the constant colour buffer"""
    torch, dev, ctx = gpu
    n = 777
    scene = synth.two_view_scene(n, seed=4)
    pair, _ = make_pair(S, gpu, scene)
    pair.estimateE(S.default_params(n, num_hypotheses=64))
    pair.computePosecandidates(); pair.choosePose(); pair.linear_triangulation()
    pts = pair.get_points()
    pos = torch.full((n, 4), -7.0, dtype=torch.float32, device=dev); vel = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    pair.copy_points_to_vbo(pos, vel, 2.0)
    torch.cuda.synchronize()
    want = np.stack([pts[0] * np.float32(2), pts[1] * np.float32(2), pts[2] * np.float32(2), np.ones(n, np.float32)], 1)
    assert same_bits(pos.cpu().numpy(), want) and (vel.cpu().numpy() == 1.0).all()
    pair.copy_points_to_vbo(pos, None)                                  # either buffer may be absent


def test_pair_reset_reuses_buffers_and_result_record(gpu):
    """sfm_pair_reset: one Image_pair serves correspondence sets of different sizes (<= its creation size) with the
    results a fresh pair gives; sfm_get_result is the individual getters in one record."""
    torch, dev, ctx = gpu
    big, small = synth.two_view_scene(3000, seed=31), synth.two_view_scene(1700, seed=32)

    def run(pair, scene, mode):
        n = len(scene["sift"])
        pair.fillXU(to_dev(torch, dev, scene["sift"]))
        pair.estimateE(S.default_params(n, num_hypotheses=700, seed=3))
        pair.computePosecandidates(mode); pair.choosePose(mode); pair.linear_triangulation(mode)
        rec = pair.get_result()
        hyp, cnt = pair.get_best()
        pind = pair.get_pose_index()
        P = (pair.get_pose_inverses() if mode == S.POSE_REFERENCE else pair.get_pose_candidates())[pind]
        assert same_bits(rec[:9], pair.get_E().reshape(9)) and same_bits(rec[9:25], P.reshape(16))
        assert tuple(int(v) for v in rec[25:28]) == (pind, cnt, hyp)
        return rec, pair.get_inlier_mask().copy(), pair.get_points().copy()

    fresh = {}
    for name, sc in (("big", big), ("small", small)):
        p = S.ImagePair(ctx, sc["K"], sc["Kinv"], 2, len(sc["sift"]))
        fresh[name] = [run(p, sc, m) for m in (S.POSE_REFERENCE, S.POSE_CORRECT)]
        p.close()
    pool = S.ImagePair(ctx, big["K"], big["Kinv"], 2, 3000)
    for name, sc in (("big", big), ("small", small), ("big", big), ("small", small)):
        pool.reset(len(sc["sift"]))
        with pytest.raises(S.SfmError):                       # state is cleared: nothing to read before fillXU / estimateE
            pool.get_result()
        for k, m in enumerate((S.POSE_REFERENCE, S.POSE_CORRECT)):
            rec, mask, pts = run(pool, sc, m)
            assert same_bits(rec, fresh[name][k][0]) and np.array_equal(mask, fresh[name][k][1]) and same_bits(pts, fresh[name][k][2])
    with pytest.raises(S.SfmError) as e:
        pool.reset(3001)
    assert e.value.code == S.E_INVALID
    with pytest.raises(S.SfmError):
        pool.reset(0)


def test_points_need_a_triangulation_of_the_current_pose(gpu):
    """sfm_get_points / sfm_copy_points_to_vbo are valid only after linear_triangulation ran for the current pose:
    not after choosePose alone, and not with the previous pair's points after sfm_pair_reset."""
    torch, dev, ctx = gpu
    n = 600
    scene = synth.two_view_scene(n, seed=31)
    pair, d_sift = make_pair(S, gpu, scene)
    pair.estimateE(S.default_params(n, num_hypotheses=64))
    pair.computePosecandidates(); pair.choosePose()
    for call in (pair.get_points, lambda: pair.copy_points_to_vbo(torch.empty(4 * n, device=dev), None)):
        with pytest.raises(S.SfmError) as e:
            call()
        assert e.value.code == S.E_STATE
    pair.linear_triangulation()
    assert np.isfinite(pair.get_points()).all()
    pair.reset(n); pair.fillXU(d_sift)
    pair.estimateE(S.default_params(n, num_hypotheses=64))
    pair.computePosecandidates(); pair.choosePose()
    with pytest.raises(S.SfmError):
        pair.get_points()
