"""GPU: the C++ facade (SfM::Image_pair / MatchSiftData with the reference's names and call order,
src/main.cpp:269-307) run as a stand-alone C++ program, compared with the oracle's chain
match -> fillXU -> estimateE -> pose candidates -> choosePose -> triangulation."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle as O
from cuda_sfm_amd_synth import synth
from helpers import same_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "cuda-sfm_amd", "host", "two_view_demo")


def make_feature_sets(n, seed):
    """Set 1 = scene points seen in image 1, set 2 = the same points in image 2, shuffled; descriptors
    make the brute-force matcher recover the correspondence (plus a few distractors)."""
    sc = synth.two_view_scene(n, seed=seed, outlier_frac=0.15)
    d1, d2, perm = synth.descriptors(n, seed=seed + 1)        # d2[i] ~ d1[perm[i]]
    s1 = synth.sift_records(d1, seed=seed + 2)
    s1["xpos"], s1["ypos"] = sc["sift"]["xpos"], sc["sift"]["ypos"]
    s2 = synth.sift_records(d2, seed=seed + 3)
    s2["xpos"], s2["ypos"] = sc["sift"]["match_xpos"][perm], sc["sift"]["match_ypos"][perm]
    extra = synth.sift_records(synth.descriptors(57, seed=seed + 4)[0], seed=seed + 5)
    return s1, np.concatenate([s2, extra]), sc


@pytest.mark.parametrize("n,H,mode,chain", [(500, 200, 0, False), (2048, 1024, 0, False), (2048, 1024, 1, False), (2048, 1024, 0, True), (777, 300, 1, True)])
def test_cpp_facade_end_to_end(tmp_path, n, H, mode, chain):
    assert os.path.exists(DEMO), "two_view_demo not built (make)"
    s1, s2, sc = make_feature_sets(n, seed=70 + n)
    f1, f2, out = (str(tmp_path / x) for x in ("s1.bin", "s2.bin", "out.bin"))
    s1.tofile(f1); s2.tofile(f2)
    env = dict(os.environ)
    if chain:
        env["SFM_DEMO_POSE_CHAIN"] = "1"                           # Image_pair::poseChain() instead of the three pose calls
    r = subprocess.run([DEMO, f1, f2, out, str(H), "0x1234", str(mode)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "MatchSiftData time" in r.stdout                      # reference prints this (matching.cu:1203)

    raw = open(out, "rb").read()
    off = 0
    def take(fmt, count):
        nonlocal off
        a = np.frombuffer(raw, dtype=fmt, count=count, offset=off); off += a.nbytes
        return a
    n_out, H_out = take("<i4", 2)
    E, P, Pinv = take("<f4", 9), take("<f4", 64), take("<f4", 64)
    pind = int(take("<i4", 1)[0]); hyp, cnt = (int(v) for v in take("<u4", 2))
    pts = take("<f4", 4 * n).reshape(4, n); mask = take("u1", n)
    rec = np.frombuffer(raw, dtype=[("score", "<f4"), ("ambiguity", "<f4"), ("match", "<i4"), ("mx", "<f4"), ("my", "<f4")], count=n, offset=off)
    assert (n_out, H_out) == (n, H)

    m = O.match_sift(s1, s2)
    assert np.array_equal(rec["match"], m["match"]) and same_bits(rec["score"], m["score"])
    assert same_bits(rec["ambiguity"], m["ambiguity"]) and same_bits(rec["mx"], m["match_xpos"]) and same_bits(rec["my"], m["match_ypos"])
    assert (m["match"][: n] < n).mean() > 0.95                     # distractors rarely win

    K, Kinv = synth.camera()
    _, _, X0, X1 = O.fill_xu(m, Kinv)
    key, counts, Ec = O.ransac_range(X0, X1, 0, H, 1e-6, 0, seed=0x1234, want_E=True)       # library default: Householder
    ocnt, ohyp = O.unpack_key(key)
    assert (hyp, cnt) == (ohyp, ocnt) and same_bits(E, Ec[ohyp])
    assert np.array_equal(mask, O.count_inliers(Ec[ohyp], X0, X1, 1e-6)[1])
    oP = O.pose_candidates(Ec[ohyp], mode)
    oind, oPinv, _, _ = O.choose_pose(X0, X1, oP, mode, 8)
    assert same_bits(P, oP.reshape(64)) and same_bits(Pinv, oPinv.reshape(64)) and pind == oind
    assert same_bits(pts, O.triangulate(X0, X1, oPinv[oind] if mode == 0 else oP[oind], 8))


def test_cpp_facade_error_is_print_and_exit(tmp_path):
    """Reference convention: print and exit non-zero (common.cu:3-15).  Fewer than 8 features cannot
    feed the 8-point solver."""
    s1, s2, _ = make_feature_sets(64, seed=5)
    f1, f2, out = (str(tmp_path / x) for x in ("s1.bin", "s2.bin", "out.bin"))
    s1[:5].tofile(f1); s2.tofile(f2)
    r = subprocess.run([DEMO, f1, f2, out, "10"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "8-point" in r.stderr


def test_cpp_homography_demo(tmp_path):
    """mainSift.cpp:72-81 re-hosted: MatchSiftData -> FindHomography -> ImproveHomography, against the oracle chain."""
    demo = os.path.join(ROOT, "cuda-sfm_amd", "host", "homography_demo")
    assert os.path.exists(demo), "homography_demo not built (make)"
    n = 1500
    hs = synth.homography_scene(n, seed=88)["sift"]
    d1, d2, perm = synth.descriptors(n, seed=89, noise=0.02, sparsity=0.8)   # d2[i] ~ d1[perm[i]]; ambiguity ~0.5
    s1 = synth.sift_records(d1, seed=90); s1["xpos"], s1["ypos"] = hs["xpos"], hs["ypos"]
    s2 = synth.sift_records(d2, seed=91); s2["xpos"], s2["ypos"] = hs["match_xpos"][perm], hs["match_ypos"][perm]
    f1, f2, out = (str(tmp_path / x) for x in ("s1.bin", "s2.bin", "h.bin"))
    s1.tofile(f1); s2.tofile(f2)
    r = subprocess.run([demo, f1, f2, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    nm, nfit = np.frombuffer(raw, "<i4", 2)
    Hr = np.frombuffer(raw, "<f4", 9, 8).reshape(3, 3); Hi = np.frombuffer(raw, "<f4", 9, 44).reshape(3, 3)
    merr = np.frombuffer(raw, "<f4", n, 80)

    m = O.match_sift(s1, s2)
    oH, onm = O.find_homography(m, 10000, 0.0, 0.80, 5.0, 0)
    assert nm == onm and same_bits(Hr, oH)
    onfit, oHi, oerr = O.improve_homography(m, oH, 5, 0.0, 0.80, 3.0)
    assert abs(int(nfit) - onfit) <= 1 and np.allclose(Hi, oHi, rtol=1e-6, atol=1e-9)
    assert np.allclose(merr, oerr, rtol=1e-3, atol=1e-3)
    assert f"Number of original features: {n} {n}" in r.stdout
    assert f"Number of matching features: {nfit} {nm} " in r.stdout
    assert (m["ambiguity"] < 0.8).mean() > 0.9                   # the demo's gate (mainSift.cpp:77) keeps most matches
    assert nm > 0.55 * n                                         # 65 % of the scene lies on the plane


def write_pgm(path, img):
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h)); f.write(img.astype(np.uint8).tobytes())


def read_sift(path):
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    return np.frombuffer(raw[4:], O.SIFT_DTYPE, n)


@pytest.mark.parametrize("args,kw", [([], dict(thresh=1.0, init_blur=1.5)),
                                     (["3.0", "1.0", "4", "1"], dict(thresh=3.0, init_blur=1.0, num_octaves=4, scale_up=True))])
def test_cpp_sift_demo(tmp_path, args, kw):
    """main.cpp:249-282 re-hosted: read two images, ExtractSift x2 (shared temp memory), MatchSiftData --
    every record of both sets against the oracle, bit for bit."""
    demo = os.path.join(ROOT, "cuda-sfm_amd", "host", "sift_demo")
    assert os.path.exists(demo), "sift_demo not built (make)"
    a = synth.image(400, 300, seed=12, blobs=150)
    b = synth.image(400, 300, seed=12, blobs=150, shift=(7.0, -4.0))           # same scene, translated
    f1, f2, o1, o2 = (str(tmp_path / x) for x in ("a.pgm", "b.pgm", "a.sift", "b.sift"))
    write_pgm(f1, a); write_pgm(f2, b)
    r = subprocess.run([demo, f1, f2, o1, o2] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Image size = (400,300)" in r.stdout and r.stdout.count("SIFT extraction time =") == 2
    assert "Incl prefiltering & memcpy =" in r.stdout and "MatchSiftData time" in r.stdout

    s1, s2 = read_sift(o1), read_sift(o2)
    e1, n1, _ = O.extract_sift(a, kw.get("num_octaves", 5), kw["init_blur"], kw["thresh"], 0.0, kw.get("scale_up", False))
    e2, n2, _ = O.extract_sift(b, kw.get("num_octaves", 5), kw["init_blur"], kw["thresh"], 0.0, kw.get("scale_up", False))
    assert (len(s1), len(s2)) == (n1, n2) and n1 > 300
    for f in ("xpos", "ypos", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data"):
        assert same_bits(s1[f], e1[f][:n1]), f
        assert same_bits(s2[f], e2[f][:n2]), f
    m = O.match_sift(e1[:n1].copy(), e2[:n2])
    for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
        assert same_bits(s1[f], m[f]), f
    # the scene moved by (7, -4): confident matches agree with that
    good = (s1["ambiguity"] < 0.8) & (s1["score"] > 0.9)
    dx, dy = s1["match_xpos"][good] - s1["xpos"][good], s1["match_ypos"][good] - s1["ypos"][good]
    assert good.sum() > 100 and abs(np.median(dx) - 7.0) < 0.5 and abs(np.median(dy) + 4.0) < 0.5


@pytest.mark.parametrize("mode", [0, 1])
def test_cpp_sfm_main_images_to_points(tmp_path, mode):
    """src/main.cpp:249-307 start to end: two images -> ExtractSift x2 -> MatchSiftData -> fillXU -> estimateE ->
    computePosecandidates -> choosePose -> linear_triangulation -> PLY.  Every stage against the oracle
    chain bit for bit, and the geometry against the synthetic scene (camera moved along +x, no rotation)."""
    app = os.path.join(ROOT, "cuda-sfm_amd", "host", "sfm_main")
    assert os.path.exists(app), "sfm_main not built (make)"
    w, h = 640, 480
    a, b, strip, disp = synth.stereo_pair(w, h, seed=5)
    f1, f2, ply, res = (str(tmp_path / x) for x in ("a.pgm", "b.pgm", "cloud.ply", "res.bin"))
    write_pgm(f1, a); write_pgm(f2, b)
    H = 2048
    r = subprocess.run([app, f1, f2, ply, res, str(H), str(mode)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr

    e1, n1, _ = O.extract_sift(a, 5, 1.5, 1.0)
    e2, n2, _ = O.extract_sift(b, 5, 1.5, 1.0)
    m = O.match_sift(e1[:n1].copy(), e2[:n2])
    K, Kinv = synth.camera(w, h)
    _, _, X0, X1 = O.fill_xu(m, Kinv)
    key, counts, Ec = O.ransac_range(X0, X1, 0, H, 1e-6, 0, seed=0x5EED5F3D, want_E=True)
    ocnt, ohyp = O.unpack_key(key)
    oP = O.pose_candidates(Ec[ohyp], mode)
    oind, oPinv, _, _ = O.choose_pose(X0, X1, oP, mode, 8)
    opts = O.triangulate(X0, X1, oPinv[oind] if mode == 0 else oP[oind], 8)

    raw = open(res, "rb").read()
    n = int(np.frombuffer(raw, "<i4", 1)[0])
    E = np.frombuffer(raw, "<f4", 9, 4); pind = int(np.frombuffer(raw, "<i4", 1, 40)[0])
    hyp, cnt = (int(v) for v in np.frombuffer(raw, "<u4", 2, 44))
    P = np.frombuffer(raw, "<f4", 16, 52).reshape(4, 4)
    pts = np.frombuffer(raw, "<f4", 4 * n, 116).reshape(4, n); mask = np.frombuffer(raw, "u1", n, 116 + 16 * n)
    assert n == n1 and (hyp, cnt) == (ohyp, ocnt) and same_bits(E, Ec[ohyp]) and pind == oind
    assert same_bits(P, oP[oind]) and same_bits(pts, opts)
    assert np.array_equal(mask, O.count_inliers(Ec[ohyp], X0, X1, 1e-6)[1])

    # geometry: x1^T E x2 = 0 with E ~ -[t]_x^T ... for t = (1, 0, 0) only E[1,2] and E[2,1] are non-zero, opposite signs
    En = E.reshape(3, 3) / np.linalg.norm(E)
    assert abs(abs(En[1, 2]) - np.sqrt(0.5)) < 0.02 and abs(abs(En[2, 1]) - np.sqrt(0.5)) < 0.02 and En[1, 2] * En[2, 1] < 0
    truth = (np.abs(m["match_xpos"] - m["xpos"] + disp[strip[np.clip(m["xpos"].astype(int), 0, w - 1)]]) < 1.0) & (np.abs(m["match_ypos"] - m["ypos"]) < 1.0)
    assert truth.mean() > 0.6 and (mask.astype(bool) & truth).sum() > 0.9 * truth.sum()      # the consensus set is the true matches
    assert (mask.astype(bool) & ~truth).sum() < 0.1 * mask.sum()
    ply_lines = open(ply).read().splitlines()
    assert ply_lines[0] == "ply" and int([l for l in ply_lines if l.startswith("element vertex")][0].split()[-1]) > 0.5 * truth.sum()
    if mode == 1:
        R = P[:3, :3]; t = P[:3, 3]
        assert np.abs(R - np.eye(3)).max() < 0.02 and abs(abs(t[0]) - 1.0) < 0.02 and np.abs(t[1:]).max() < 0.05
        # depth is inversely proportional to disparity: facets with larger disparity are closer
        ok = mask.astype(bool) & truth & (pts[3] != 0)
        z = pts[2, ok] / pts[3, ok]
        d = disp[strip[np.clip(m["xpos"][ok].astype(int), 0, w - 1)]]
        assert (z > 0).mean() > 0.98 or (z < 0).mean() > 0.98
        zd = np.abs(z) * d                                                     # = f * |t| = const
        assert np.std(zd) / np.mean(zd) < 0.05
