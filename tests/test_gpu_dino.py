"""BASELINE configs[1] on the REAL data: the two frames the reference program reads (src/main.cpp:250-251,
data/dino/viff.000.ppm / viff.001.ppm, kept as 8-bit grey fixtures under tests/golden/dino/ -- the conversion
cv::imread(path, 0) applies), its own parameters
(initBlur 1.5, thresh 1.0, 5 octaves, 32768 points, K of main.cpp:292-297, H = N/8, threshold 1e-6):
ExtractSift x2 -> MatchSiftData -> fillXU -> estimateE -> computePosecandidates -> choosePose ->
linear_triangulation, every stage bit for bit against the oracle chain and the frozen oracle results; then the
4-view ring (configs[4] in miniature) and the re-hosted main program on the same files."""
import os
import subprocess

import numpy as np
import pytest

import cuda_sfm_amd as S
import oracle as O
from helpers import DINO_K, DINO_KINV, DINO_SIFT, read_pnm_grey, same_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DINO = os.path.join(ROOT, "tests", "golden", "dino")
FIELDS = ("xpos", "ypos", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data")


def frame(k):
    return os.path.join(DINO, f"dino_grey_{k:03d}.pgm")


def extract(gpu, img, max_pts=32768):
    torch, dev, ctx = gpu
    h, w = img.shape
    pitch = (w + 127) // 128 * 128
    pad = np.zeros((h, pitch), np.float32); pad[:, :w] = img
    d_sift = torch.zeros((max_pts, 576), dtype=torch.uint8, device=dev)
    n, stored = ctx.extract_sift(d_sift, max_pts, torch.from_numpy(pad).to(dev), w, h, pitch, **DINO_SIFT)
    return d_sift, n, stored


def oracle_features(img):
    return O.extract_sift(img, DINO_SIFT["num_octaves"], DINO_SIFT["init_blur"], DINO_SIFT["thresh"], 0.0, False, 32768)


def test_dino_pair_every_stage(gpu):
    torch, dev, ctx = gpu
    g = np.load(os.path.join(ROOT, "tests", "golden", "dino_oracle.npz"))
    imgs = [read_pnm_grey(frame(k)) for k in (0, 1)]
    assert imgs[0].shape == (576, 720)
    d, n, st = zip(*[extract(gpu, im) for im in imgs])
    of = [oracle_features(im) for im in imgs]
    assert list(n) == [f[1] for f in of] == g["num_pts"].tolist() and list(st) == [f[2] for f in of]
    for k in (0, 1):
        rec = d[k].cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)
        for f in FIELDS:
            assert same_bits(rec[f][:st[k]], of[k][0][f][:st[k]]), (k, f)
    n1, n2 = n
    ctx.match(d[0], n1, d[1], n2)                                              # MatchSiftData
    m = d[0].cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)[:n1]
    om = O.match_sift(of[0][0][:n1].copy(), of[1][0][:n2])
    assert np.array_equal(m["match"], om["match"]) and np.array_equal(m["match"], g["match"])
    for f in ("score", "ambiguity", "match_xpos", "match_ypos"):
        assert same_bits(m[f], om[f]), f

    pair = S.ImagePair(ctx, DINO_K, DINO_KINV, 2, n1)                          # main.cpp:298-307
    pair.fillXU(d[0])
    p = S.default_params(n1)
    assert p.num_hypotheses == n1 // 8
    pair.estimateE(p)
    _, _, X0, X1 = O.fill_xu(om, DINO_KINV)
    key, ocounts, oE = O.ransac_range(X0, X1, 0, p.num_hypotheses, p.threshold, p.jacobi_sweeps, seed=p.seed, want_E=True)
    ocnt, ohyp = O.unpack_key(key)
    assert np.array_equal(pair.get_inlier_counts(p.num_hypotheses), ocounts) and np.array_equal(ocounts, g["counts"])
    assert pair.get_best() == (ohyp, ocnt) == tuple(int(v) for v in g["best"])
    assert same_bits(pair.get_E(), oE[ohyp].reshape(3, 3)) and same_bits(oE[ohyp], g["E"])
    assert ocnt > 0.25 * n1                                                    # a real consensus set (594 of 2155)
    assert np.array_equal(pair.get_inlier_mask(), O.count_inliers(oE[ohyp], X0, X1, p.threshold)[1])
    for mode in (S.POSE_REFERENCE, S.POSE_CORRECT):
        pair.computePosecandidates(mode); pair.choosePose(mode); pair.linear_triangulation(mode)
        oP = O.pose_candidates(oE[ohyp], mode)
        oind, oPinv, _, _ = O.choose_pose(X0, X1, oP, mode, 8)
        assert pair.get_pose_index() == oind and same_bits(pair.get_pose_candidates(), oP)
        opts = O.triangulate(X0, X1, oPinv[oind] if mode == S.POSE_REFERENCE else oP[oind], 8)
        assert same_bits(pair.get_points(), opts)


def test_dino_ring_of_four_views(gpu):
    """configs[4] in miniature on real frames 0..3: process_views (extract per view, ring pairs, per-pair pipeline)."""
    torch, dev, ctx = gpu
    views = [read_pnm_grey(frame(k)) for k in range(4)]
    res, counts = S.process_views(ctx, views, DINO_K, DINO_KINV, max_pts=32768, sift=DINO_SIFT, device=dev)
    feats = [oracle_features(v) for v in views]
    assert counts == [f[1] for f in feats] and min(counts) > 1500
    for pid, (i, j) in enumerate(S.ring_pairs(4)):
        ni = feats[i][1]
        m = O.match_sift(feats[i][0][:ni].copy(), feats[j][0][:feats[j][1]])
        _, _, X0, X1 = O.fill_xu(m, DINO_KINV)
        p = S.default_params(ni)
        key, _, Ec = O.ransac_range(X0, X1, 0, p.num_hypotheses, p.threshold, p.jacobi_sweeps, seed=p.seed, want_E=True)
        ocnt, ohyp = O.unpack_key(key)
        r = res[pid]
        assert same_bits(r[:9], Ec[ohyp]) and (int(r[26]), int(r[27])) == (ocnt, ohyp)


def test_dino_ring_of_36_views(gpu):
    """BASELINE configs[4] on the reference's own data: all 36 frames of data/dino (8-bit grey fixtures), ExtractSift per
    view, the 36 ring pairs through sfm_extract_views + sfm_process_pairs.  Every view's feature count against the oracle
    extractor, six pairs (among them the closing pair 35-0) against the whole oracle chain, every pair a valid estimate."""
    torch, dev, ctx = gpu
    from helpers import dino_frame
    views = [read_pnm_grey(dino_frame(k)) for k in range(36)]
    res, counts = S.process_views(ctx, views, DINO_K, DINO_KINV, max_pts=32768, sift=DINO_SIFT, device=dev)
    assert sorted(res) == list(range(36)) and min(counts) > 1000
    ring = S.ring_pairs(36)
    check = [0, 1, 7, 18, 29, 35]
    need = sorted({v for pid in check for v in ring[pid]})
    feats = {v: oracle_features(views[v]) for v in need}
    for v in need:
        assert counts[v] == feats[v][1], f"view {v}"
    for pid in check:
        i, j = ring[pid]
        ni = feats[i][1]
        m = O.match_sift(feats[i][0][:ni].copy(), feats[j][0][:feats[j][1]])
        _, _, X0, X1 = O.fill_xu(m, DINO_KINV)
        p = S.default_params(ni)
        key, _, Ec = O.ransac_range(X0, X1, 0, p.num_hypotheses, p.threshold, p.jacobi_sweeps, seed=p.seed, want_E=True)
        ocnt, ohyp = O.unpack_key(key)
        r = res[pid]
        assert same_bits(r[:9], Ec[ohyp]) and (int(r[26]), int(r[27])) == (ocnt, ohyp), f"pair {pid}"
        oP = O.pose_candidates(Ec[ohyp], S.POSE_REFERENCE)
        oind, oPinv, _, _ = O.choose_pose(X0, X1, oP, S.POSE_REFERENCE, 8)
        assert int(r[25]) == oind and same_bits(r[9:25], oPinv[oind].reshape(16))
    for pid in range(36):
        assert res[pid][26] >= 8 and np.isfinite(res[pid][:25]).all()


def test_dino_all_630_pairs_every_record_equals_the_oracle_chain(gpu):
    """BASELINE configs[4], all 630 unordered pairs of the 36 dino frames, SWEPT: every pair's record out of the batched
    sfm_process_pairs (many-matches launches, five launches for the rest of the chain) against the oracle chain on the same
    features -- MatchC1 restatement (CudaSift/match.cu:57-71) -> fillXU -> estimateE (H = n/8) -> computePosecandidates ->
    choosePose (src/main.cpp:282-307 per pair): best hypothesis, inlier count, E bit for bit, pose index, the chosen pose.
    The match itself is compared pair by pair as well: sfm_match's index / score / ambiguity / match position fields against
    the oracle's for all 630 pairs (the index arrays the batched launch feeds fillXU with are these)."""
    torch, dev, ctx = gpu
    from helpers import dino_frame
    views = [read_pnm_grey(dino_frame(k)) for k in range(36)]
    pairs = [(i, j) for i in range(36) for j in range(i + 1, 36)]
    res, counts = S.process_views(ctx, views, DINO_K, DINO_KINV, pairs=pairs, max_pts=8192, sift=DINO_SIFT, device=dev)
    assert ctx.last_pairs_batched() and sorted(res) == list(range(630))
    feats = []
    for v in views:                                            # the features the extractor delivers (its own parity: the tests above)
        d, n, st = extract(gpu, v, 8192)
        feats.append((d, n, d.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)[:n].copy()))
    assert counts == [f[1] for f in feats]
    bad = []
    for pid, (i, j) in enumerate(pairs):
        (di, ni, fi), (dj, nj, fj) = feats[i], feats[j]
        om = O.match_sift(fi.copy(), fj)
        ctx.match(di, ni, dj, nj)                              # writes view i's match fields (di is re-used: fillXU reads only positions)
        gm = di.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)[:ni]
        ok = np.array_equal(gm["match"], om["match"]) and all(same_bits(gm[f], om[f]) for f in ("score", "ambiguity", "match_xpos", "match_ypos"))
        _, _, X0, X1 = O.fill_xu(om, DINO_KINV)
        p = S.default_params(ni)
        key, _, Ec = O.ransac_range(X0, X1, 0, p.num_hypotheses, p.threshold, p.jacobi_sweeps, seed=p.seed, want_E=True)
        ocnt, ohyp = O.unpack_key(key)
        oP = O.pose_candidates(Ec[ohyp], S.POSE_REFERENCE)
        oind, oPinv, _, _ = O.choose_pose(X0, X1, oP, S.POSE_REFERENCE, 8)
        r = res[pid]
        ok = ok and same_bits(r[:9], Ec[ohyp]) and (int(r[26]), int(r[27])) == (ocnt, ohyp) and int(r[25]) == oind \
            and same_bits(r[9:25], oPinv[oind].reshape(16))
        if not ok:
            bad.append((pid, i, j))
    assert not bad, f"{len(bad)} of 630 pairs differ from the oracle chain, first: {bad[:5]}"


def test_dino_extract_views_equals_single_extractions(gpu):
    """sfm_extract_views against sfm_extract_sift view by view: feature counts and every byte of the records, twelve frames,
    float and 8-bit images (eight per-view chains on eight streams, one worker thread each)."""
    torch, dev, ctx = gpu
    import ctypes as C
    from helpers import dino_frame
    views = [read_pnm_grey(dino_frame(k)) for k in range(12)]
    h, w = views[0].shape
    p = (w + 127) // 128 * 128
    max_pts = 8192
    single = []
    for v in views:
        pad = np.zeros((h, p), np.float32); pad[:, :w] = v
        d_sift = torch.zeros((max_pts, 576), dtype=torch.uint8, device=dev)
        n, _ = ctx.extract_sift(d_sift, max_pts, torch.from_numpy(pad).to(dev), w, h, p, **DINO_SIFT)
        single.append((n, d_sift.cpu().numpy()[:n].copy()))
    for u8 in (False, True):
        imgs = [np.ascontiguousarray(v, np.uint8 if u8 else np.float32) for v in views]
        ct = C.c_ubyte if u8 else C.c_float
        ptrs = (C.POINTER(ct) * len(imgs))(*[im.ctypes.data_as(C.POINTER(ct)) for im in imgs])
        block = torch.zeros((len(imgs), max_pts * 576 + 64), dtype=torch.uint8, device=dev)
        cnts = (C.c_int * len(imgs))()
        fn = S._lib.sfm_extract_views_u8 if u8 else S._lib.sfm_extract_views
        rc = fn(ctx._h, ptrs, len(imgs), w, h, 0, 1, block.data_ptr(), max_pts * 576 + 64, max_pts, int(DINO_SIFT["num_octaves"]),
                float(DINO_SIFT["init_blur"]), float(DINO_SIFT["thresh"]), float(DINO_SIFT.get("lowest_scale", 0.0)), 0, cnts)
        assert rc == 0, S.last_error() if hasattr(S, "last_error") else rc
        got = block.cpu().numpy()
        for k, (n, rec) in enumerate(single):
            assert cnts[k] == n, f"view {k} ({'8-bit' if u8 else 'float'}): {cnts[k]} features, single extraction {n}"
            assert int(got[k, max_pts * 576:max_pts * 576 + 4].view(np.int32)[0]) == n
            assert np.array_equal(got[k, :n * 576].reshape(n, 576), rec), f"view {k}: records differ"


def test_dino_views_as_8_bit_images_give_the_same_records(gpu):
    """sfm_extract_views_u8 (8-bit grey images: a quarter of the PCIe bytes, widened on the device) against the float entry
    point: feature counts and every result record of the ring of six frames, bit for bit."""
    torch, dev, ctx = gpu
    from helpers import dino_frame
    views = [read_pnm_grey(dino_frame(k)) for k in range(6)]
    bytes_ = [v.astype(np.uint8) for v in views]
    assert all(np.array_equal(b.astype(np.float32), v) for b, v in zip(bytes_, views))       # the fixtures ARE 8-bit
    res_f, counts_f = S.process_views(ctx, views, DINO_K, DINO_KINV, max_pts=8192, sift=DINO_SIFT, device=dev)
    res_b, counts_b = S.process_views(ctx, bytes_, DINO_K, DINO_KINV, max_pts=8192, sift=DINO_SIFT, device=dev)
    assert counts_b == counts_f and sorted(res_b) == sorted(res_f) == list(range(6))
    for pid in res_f:
        assert same_bits(res_b[pid], res_f[pid]), f"pair {pid}"


def test_dino_ring_batched_equals_per_pair(gpu):
    """sfm_process_pairs (every pair enqueued back to back inside the C library, records assembled on the device, one
    read-back) against the same pairs taken one at a time through the Image_pair calls: records bit for bit, for both
    pose modes, a second pass over the pooled Image_pair included; ranks 0 / 1 of 2 split the list without overlap."""
    torch, dev, ctx = gpu
    views = [read_pnm_grey(frame(k)) for k in range(4)]
    h, w = views[0].shape
    p = (w + 127) // 128 * 128
    feats = []
    for v in views:
        pad = np.zeros((h, p), np.float32); pad[:, :w] = v
        d_sift = torch.zeros((32768, 576), dtype=torch.uint8, device=dev)
        n, _ = ctx.extract_sift(d_sift, 32768, torch.from_numpy(pad).to(dev), w, h, p, **DINO_SIFT)
        feats.append((d_sift, n))
    ring = S.ring_pairs(4)
    for mode in (S.POSE_REFERENCE, S.POSE_CORRECT):
        single = []
        for (i, j) in ring:                                         # one at a time: match, then the Image_pair sequence
            (s1, n1), (s2, n2) = feats[i], feats[j]
            ctx.match(s1, n1, s2, n2)
            pair = S.ImagePair(ctx, DINO_K, DINO_KINV, 2, n1)
            pair.fillXU(s1); pair.estimateE(S.default_params(n1))
            pair.computePosecandidates(mode); pair.choosePose(mode); pair.linear_triangulation(mode)
            single.append(pair.get_result().copy())
            pair.close()
        descs = [(feats[i][0], feats[i][1], feats[j][0], feats[j][1]) for (i, j) in ring]
        for _ in range(2):
            rec, status = S.process_pairs_local(ctx, descs, DINO_K, DINO_KINV, pose_mode=mode)
            assert status.tolist() == [0, 0, 0, 0]
            for k in range(4):
                assert same_bits(rec[k], single[k]), f"pair {k} mode {mode}"
        r0, _ = S.process_pairs_local(ctx, descs, DINO_K, DINO_KINV, rank=0, world=2, pose_mode=mode)
        r1, _ = S.process_pairs_local(ctx, descs, DINO_K, DINO_KINV, rank=1, world=2, pose_mode=mode)
        assert same_bits(r0[0], single[0]) and same_bits(r0[1], single[2]) and same_bits(r1[0], single[1]) and same_bits(r1[1], single[3])
    # a pair with too few features is reported, not run
    rec, status = S.process_pairs_local(ctx, [(feats[0][0], 5, feats[1][0], feats[1][1]), descs[0]], DINO_K, DINO_KINV, pose_mode=S.POSE_CORRECT)
    assert status.tolist() == [S.E_INVALID, 0] and (rec[0] == -1).all() and same_bits(rec[1], single[0])


def test_dino_main_program(tmp_path):
    """host/sfm_main (src/main.cpp:249-307 re-hosted) on the reference's own two files with its own defaults."""
    app = os.path.join(ROOT, "cuda-sfm_amd", "host", "sfm_main")
    assert os.path.exists(app), "sfm_main not built (make)"
    ply, res = str(tmp_path / "dino.ply"), str(tmp_path / "dino.bin")
    r = subprocess.run([app, frame(0), frame(1), ply, res], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    g = np.load(os.path.join(ROOT, "tests", "golden", "dino_oracle.npz"))
    raw = open(res, "rb").read()
    n = int(np.frombuffer(raw, "<i4", 1)[0])
    E = np.frombuffer(raw, "<f4", 9, 4)
    hyp, cnt = (int(v) for v in np.frombuffer(raw, "<u4", 2, 44))
    assert n == int(g["num_pts"][0]) and (hyp, cnt) == tuple(int(v) for v in g["best"]) and same_bits(E, g["E"])
    lines = open(ply).read().splitlines()
    assert lines[0] == "ply" and int([l for l in lines if l.startswith("element vertex")][0].split()[-1]) > 100


def test_dino_pair_with_1024_hypotheses(gpu):
    """BASELINE configs[1] as written: the dino pair, ~2k matches, 1024 RANSAC hypotheses (the reference's own run uses
    H = N/8 = 269: test_dino_pair_every_stage).  Match + estimateE end to end on the real frames, every count, the key, E and
    the mask against the oracle chain fed with the GPU's features and matches."""
    torch, dev, ctx = gpu
    imgs = [read_pnm_grey(frame(k)) for k in (0, 1)]
    (d1, n1, _), (d2, n2, _) = extract(gpu, imgs[0]), extract(gpu, imgs[1])
    ctx.match(d1, n1, d2, n2)
    m = d1.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)[:n1]
    f2 = d2.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)[:n2]
    om = O.match_sift(m.copy(), f2)
    assert np.array_equal(m["match"], om["match"]) and same_bits(m["score"], om["score"])
    pair = S.ImagePair(ctx, DINO_K, DINO_KINV, 2, n1)
    pair.fillXU(d1)
    H = 1024
    p = S.default_params(n1, num_hypotheses=H)
    pair.estimateE(p)
    _, _, X0, X1 = O.fill_xu(om, DINO_KINV)
    key, ocounts, oE = O.ransac_range(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed, want_E=True)
    ocnt, ohyp = O.unpack_key(key)
    assert np.array_equal(pair.get_inlier_counts(H), ocounts) and pair.get_key() == key
    assert pair.get_best() == (ohyp, ocnt) and same_bits(pair.get_E(), oE[ohyp].reshape(3, 3))
    assert np.array_equal(pair.get_inlier_mask(), O.count_inliers(oE[ohyp], X0, X1, p.threshold)[1])
    assert ocnt >= 594                                       # at least the consensus set the 269-hypothesis run finds
