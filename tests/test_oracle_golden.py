"""CPU: the oracle against the committed golden vectors.
svd3_ref / match_ref hold outputs of the REFERENCE's own code (SfM/svd.h, CudaSift/match.cu MatchC1)
compiled in place by oracle/ref_build.sh and recorded by tests/gen_golden.py."""
import os

import numpy as np

import oracle as O
from helpers import same_bits

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_svd3_bit_exact_vs_reference_svd_h():
    g = np.load(os.path.join(G, "svd3_ref.npz"))
    for i, a in enumerate(g["A"]):
        u, s, v = O.svd3(a)
        assert same_bits(u.reshape(9), g["U"][i]) and same_bits(s.reshape(9), g["S"][i]) and same_bits(v.reshape(9), g["V"][i]), i
        assert same_bits(O.normalizeE(a).reshape(9), g["normalizeE"][i]), i
        d = np.float32(O.det_ref(a))
        assert same_bits(np.array([d]), g["det"][i:i + 1]), i
        b = g["B"][i]
        assert same_bits(O.multAB(a, b).reshape(9), g["AB"][i])
        assert same_bits(O.multAtB(a, b).reshape(9), g["AtB"][i])
        assert same_bits(O.multABt(a, b).reshape(9), g["ABt"][i])


def test_pose_sign_fix_matches_reference_host_code():
    """computePosecandidates' host part (sfm.cu:240-245): svd, det(u v^T) as written, neg(v)."""
    g = np.load(os.path.join(G, "svd3_ref.npz"))
    for i, a in enumerate(g["A"]):
        u, s, v = O.svd3(a)
        if O.det_ref(O.multABt(u, v)) < 0:
            v = -v
        assert same_bits(u.reshape(9), g["pose_u"][i]) and same_bits(v.reshape(9), g["pose_v"][i]), i
        if np.isfinite(u).all() and np.isfinite(v).all():
            P = O.pose_candidates(a, O.POSE_REFERENCE)
            # candidate translations are -/+ the third column of u (kernels.h:368-370)
            assert same_bits(P[1, :3, 3], u[:, 2]) and same_bits(P[0, :3, 3], -u[:, 2])


def test_svd3_is_a_decomposition():
    g = np.load(os.path.join(G, "svd3_ref.npz"))
    for a in g["A"][:200]:
        u, s, v = O.svd3(a)
        if not np.isfinite(u).all():
            continue
        rec = u.astype(np.float64) @ s.astype(np.float64) @ v.astype(np.float64).T
        assert np.abs(rec - a.reshape(3, 3)).max() <= 2e-5 * max(1.0, np.abs(a).max())


def test_match_vs_reference_matchC1():
    g = np.load(os.path.join(G, "match_ref.npz"))
    best, second, idx = O.match_desc(g["d1"], g["d2"])
    # bit-exact against MatchC1 built with FMA contraction (what nvcc does to matching.cu:338-351)
    assert np.array_equal(idx, g["index_fma"]) and same_bits(best, g["score_fma"])
    # index-exact against the plain build; its scores differ only by fused-vs-unfused rounding
    assert np.array_equal(idx, g["index_plain"])
    assert np.abs(best - g["score_plain"]).max() <= 4 * 128 * np.finfo(np.float32).eps
    assert (second <= best).all()
    assert (g["perm"][idx] == np.arange(len(idx))).mean() > 0.99


def test_e2e_regression_vectors():
    g = np.load(os.path.join(G, "e2e_oracle.npz"))
    sift = np.zeros(len(g["xpos"]), O.SIFT_DTYPE)
    sift["xpos"], sift["ypos"], sift["match_xpos"], sift["match_ypos"] = g["xpos"], g["ypos"], g["mxpos"], g["mypos"]
    _, _, X0, X1 = O.fill_xu(sift, g["Kinv"])
    assert same_bits(X0, g["X0"]) and same_bits(X1, g["X1"])
    H = len(g["counts"])
    key, counts, Ec = O.ransac_range(X0, X1, 0, H, 1e-6, 7, seed=42, want_E=True)
    assert np.array_equal(counts, g["counts"]) and key == int(g["key"]) and same_bits(Ec, g["Ecand"])
    assert np.array_equal(np.array([O.sample8(42, h, len(sift)) for h in range(H)]), g["idx"])
    cnt, hyp = O.unpack_key(key)
    assert hyp == int(np.argmax(counts)) and cnt == counts.max()          # first maximum
    _, mask = O.count_inliers(Ec[hyp], X0, X1, 1e-6)
    assert np.array_equal(mask, g["mask"])
    keyq, countsq, Ecq = O.ransac_range(X0, X1, 0, H, 1e-6, 0, seed=42, want_E=True)      # Householder solver
    assert np.array_equal(countsq, g["qr_counts"]) and keyq == int(g["qr_key"]) and same_bits(Ecq, g["qr_Ecand"])
    assert np.array_equal(O.count_inliers(Ecq[O.unpack_key(keyq)[1]], X0, X1, 1e-6)[1], g["qr_mask"])
    for mode in (0, 1):
        P = O.pose_candidates(Ec[hyp], mode)
        ind, Pinv, _, _ = O.choose_pose(X0, X1, P, mode, 8)
        assert same_bits(P, g[f"P{mode}"]) and ind == int(g[f"pind{mode}"]) and same_bits(Pinv, g[f"Pinv{mode}"])
        pts = O.triangulate(X0, X1, Pinv[ind] if mode == 0 else P[ind], 8)
        assert same_bits(pts, g[f"points{mode}"])


def test_reference_gpu_kernels_golden():
    """Outputs of the REFERENCE'S OWN CUDA kernels (cudaSiftD.cu LowPassBlock / ScaleDown / ScaleUp /
    LaplaceMultiMem / FindPointsMulti, matching.cu ComputeHomographies / TestHomographies) compiled for gfx950 in place and
    run on an MI355X (tests/gen_golden_gpu.py) -- the oracle reproduces them bit for bit on the CPU."""
    g = np.load(os.path.join(G, "ref_gpu_kernels.npz"))
    img = g["sift_image"]
    # the filter tables are host code of the reference, restated in the oracle; the fixture stores what was used
    kt, k5 = O.sift_tables(5)
    assert same_bits(kt, g["sift_laplace_table"]) and same_bits(k5, g["sift_scaledown_taps"])
    for tag, blur in (("lp10", 1.0), ("lp15", 1.5)):
        assert same_bits(O.sift_lowpass_taps(blur), g["sift_" + tag + "_taps"])
        assert same_bits(O.sift_lowpass(img, g["sift_" + tag + "_taps"]), g["sift_" + tag])
    assert same_bits(O.sift_scaledown(img, k5), g["sift_scaledown"])
    assert same_bits(O.sift_scaleup(img), g["sift_scaleup"])
    for octave in (5, 2):
        assert same_bits(O.sift_laplace(g["sift_lp10"], kt.reshape(8, 192)[octave][:128]), g[f"sift_dog_octave{octave}"])
    # FindPointsMulti: same point set, fields bit for bit (scale: device powf / exp2f, 4e-7)
    fpts, fcnt = O.sift_find_points(g["sift_dog_octave5"], 1.0, 0.0, float(g["find_thresh"]), 4096)
    assert fcnt == len(g["find_xpos"]) and fcnt > 50
    fpts = fpts[np.lexsort((fpts["scale"], fpts["xpos"], fpts["ypos"]))]
    for f in ("xpos", "ypos", "sharpness", "edgeness"):
        assert same_bits(fpts[f], g["find_" + f]), f
    assert np.abs(fpts["scale"] / g["find_scale"] - 1.0).max() < 4e-7
    coord, pts = g["homo_coord"], g["homo_pts"]
    thr = np.float32(g["homo_thresh"])
    for l in range(pts.shape[1]):
        h = O.homography4(coord, pts[:, l])
        assert same_bits(h, g["homo_h"][:, l].copy())
        assert O.homography_count(h, coord, coord.shape[1], thr * thr) == g["homo_counts"][l]


def test_dino_pair_oracle_results_are_frozen():
    """The oracle chain on the reference program's own input pair (tests/golden/dino, src/main.cpp:250-307 parameters)
    reproduces the committed results: features, matches, per-hypothesis counts, winner, E."""
    from helpers import DINO_KINV, DINO_SIFT, read_pnm_grey
    gold = np.load(os.path.join(G, "dino_oracle.npz"))
    imgs = [read_pnm_grey(os.path.join(G, "dino", f"dino_grey_{k:03d}.pgm")) for k in (0, 1)]
    feats = [O.extract_sift(im, DINO_SIFT["num_octaves"], DINO_SIFT["init_blur"], DINO_SIFT["thresh"], 0.0, False, 32768) for im in imgs]
    assert [f[1] for f in feats] == gold["num_pts"].tolist() and [f[2] for f in feats] == gold["stored"].tolist()
    n1 = feats[0][1]
    for f, k in (("xpos", "xpos0"), ("ypos", "ypos0"), ("scale", "scale0"), ("orientation", "orientation0")):
        assert same_bits(feats[0][0][f][:n1], gold[k]), f
    assert same_bits(feats[0][0]["data"][:32], gold["desc0_head"])
    m = O.match_sift(feats[0][0][:n1].copy(), feats[1][0][:feats[1][1]])
    assert np.array_equal(m["match"], gold["match"]) and same_bits(m["score"], gold["score"]) and same_bits(m["ambiguity"], gold["ambiguity"])
    _, _, X0, X1 = O.fill_xu(m, DINO_KINV)
    key, counts, Ec = O.ransac_range(X0, X1, 0, n1 // 8, 1e-6, 0, seed=0x5EED5F3D, want_E=True)
    cnt, hyp = O.unpack_key(key)
    assert np.array_equal(counts, gold["counts"]) and [hyp, cnt] == gold["best"].tolist() and same_bits(Ec[hyp], gold["E"])
