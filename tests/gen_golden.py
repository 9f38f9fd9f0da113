#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container, where
/root/reference exists and `make -C oracle` has produced oracle/_ref/*.so).

  svd3_ref.npz     inputs + outputs of the REFERENCE's own SfM/svd.h functions (svd, multAB/AtB/ABt,
                   det as written, normalizeE body, computePosecandidates sign fix), compiled in
                   place -> pins the oracle's 3x3 algebra bit for bit.
  match_ref.npz    512 x 512 x 128 descriptors + the REFERENCE's CPU matcher MatchC1
                   (CudaSift/match.cu:57-71) built plain and with FMA contraction.
  e2e_oracle.npz   seeded two-view scene -> oracle E / counts / mask / poses / points.  These are
                   ORACLE outputs (regression vectors for the HIP path and for oracle refactors),
                   not reference outputs: the reference's estimateE is not reproducible.
  dino/, dino_oracle.npz   the reference program's own input images (data/dino/viff.000 .. 003.ppm, stored as 8-bit grey
                   PGM) and the oracle's results on that pair (ORACLE outputs, see gen_dino).
Only data is written; no reference source text is stored."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
from cuda_sfm_amd_synth import synth

OUT = os.path.join(HERE, "golden")
f32p = O.f32p


def fp(a):
    return a.ctypes.data_as(f32p)


def gen_svd3():
    R = O.ref_lib("libref_svd.so")
    rng = np.random.default_rng(20261001)
    mats = []
    for t in range(256):
        k = t % 8
        if k == 0: a = rng.standard_normal(9)
        elif k == 1: a = rng.standard_normal(9) * 10 ** rng.uniform(-5, 5)
        elif k == 2: a = np.outer(rng.standard_normal(3), rng.standard_normal(3)).reshape(9)          # rank 1
        elif k == 3: a = (np.outer(rng.standard_normal(3), rng.standard_normal(3)) + np.outer(rng.standard_normal(3), rng.standard_normal(3))).reshape(9)  # rank 2
        elif k == 4: a = np.linalg.qr(rng.standard_normal((3, 3)))[0].reshape(9)                      # orthogonal
        elif k == 5: a = np.diag(rng.standard_normal(3)).reshape(9)
        elif k == 6:
            a = rng.standard_normal(9); a[rng.integers(0, 9, 4)] = 0
        else:
            x = rng.standard_normal(3); a = np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]]).reshape(9)  # skew (essential-like)
        mats.append(a)
    mats += [np.zeros(9), np.eye(3).reshape(9), -np.eye(3).reshape(9)]
    A = np.ascontiguousarray(np.array(mats), np.float32)
    n = len(A)
    U = np.empty_like(A); S = np.empty_like(A); V = np.empty_like(A)
    NE = A.copy(); PU = np.empty_like(A); PV = np.empty_like(A)
    DET = np.empty(n, np.float32)
    B = np.ascontiguousarray(rng.standard_normal((n, 9)), np.float32)
    AB = np.empty_like(A); AtB = np.empty_like(A); ABt = np.empty_like(A)
    for i in range(n):
        R.ref_svd3(fp(A[i]), fp(U[i]), fp(S[i]), fp(V[i]))
        R.ref_normalizeE(fp(NE[i]))
        R.ref_pose_uv(fp(A[i]), fp(PU[i]), fp(PV[i]))
        DET[i] = R.ref_det(fp(A[i]))
        R.ref_multAB(fp(A[i]), fp(B[i]), fp(AB[i]))
        R.ref_multAtB(fp(A[i]), fp(B[i]), fp(AtB[i]))
        R.ref_multABt(fp(A[i]), fp(B[i]), fp(ABt[i]))
    np.savez_compressed(os.path.join(OUT, "svd3_ref.npz"), A=A, B=B, U=U, S=S, V=V, normalizeE=NE,
                        pose_u=PU, pose_v=PV, det=DET, AB=AB, AtB=AtB, ABt=ABt)
    print("svd3_ref.npz", n, "matrices")


def gen_match():
    n = 512
    d1, d2, perm = synth.descriptors(n, seed=7)
    out = {"d1": d1, "d2": d2, "perm": perm.astype(np.int32)}
    for lib, tag in (("libref_match.so", "plain"), ("libref_match_fma.so", "fma")):
        M = O.ref_lib(lib)
        sc = np.zeros(n, np.float32); ix = np.zeros(n, np.int32)
        M.ref_matchC1(n, fp(d1), fp(d2), fp(sc), ix.ctypes.data_as(O.i32p))
        out["score_" + tag] = sc; out["index_" + tag] = ix
    np.savez_compressed(os.path.join(OUT, "match_ref.npz"), **out)
    print("match_ref.npz", (out["index_plain"] != out["index_fma"]).sum(), "index diffs plain vs fma")


def gen_e2e():
    n, H = 600, 160
    sc = synth.two_view_scene(n, seed=1234)
    U0, U1, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    key, counts, Ec = O.ransac_range(X0, X1, 0, H, 1e-6, 7, seed=42, want_E=True)
    cnt, hyp = O.unpack_key(key)
    _, mask = O.count_inliers(Ec[hyp], X0, X1, 1e-6)
    out = {"xpos": sc["sift"]["xpos"], "ypos": sc["sift"]["ypos"], "mxpos": sc["sift"]["match_xpos"],
           "mypos": sc["sift"]["match_ypos"], "K": sc["K"], "Kinv": sc["Kinv"], "X0": X0, "X1": X1,
           "counts": counts, "Ecand": Ec, "key": np.uint64(key), "mask": mask,
           "idx": np.array([O.sample8(42, h, n) for h in range(H)], np.int32)}
    for mode in (0, 1):
        P = O.pose_candidates(Ec[hyp], mode)
        ind, Pinv, d1, d2 = O.choose_pose(X0, X1, P, mode, 8)
        pts = O.triangulate(X0, X1, Pinv[ind] if mode == 0 else P[ind], 8)
        out.update({f"P{mode}": P, f"Pinv{mode}": Pinv, f"pind{mode}": np.int32(ind), f"points{mode}": pts})
    # the same scene through the Householder null-vector solver (jacobi_sweeps = 0, the library default)
    keyq, countsq, Ecq = O.ransac_range(X0, X1, 0, H, 1e-6, 0, seed=42, want_E=True)
    cq, hq = O.unpack_key(keyq)
    out.update(qr_counts=countsq, qr_Ecand=Ecq, qr_key=np.uint64(keyq), qr_mask=O.count_inliers(Ecq[hq], X0, X1, 1e-6)[1])
    np.savez_compressed(os.path.join(OUT, "e2e_oracle.npz"), **out)
    print("e2e_oracle.npz best", hyp, cnt, "householder best", hq, cq)


def gen_dino():
    """The reference program's own input (src/main.cpp:250-251 reads data/dino/viff.000.ppm and viff.001.ppm):
    all 36 frames are kept as 8-bit grey PGM data fixtures (0 and 1: the pair of main.cpp, 0..3: the 4-view ring, 0..35: the ring of BASELINE configs[4]), and the
    oracle's results on the pair are frozen so that a drift of the oracle shows up in the CPU suite."""
    from helpers import read_pnm_grey, DINO_KINV, DINO_SIFT
    src = "/root/reference/data/dino"
    dst = os.path.join(OUT, "dino"); os.makedirs(dst, exist_ok=True)
    # stored as 8-bit grey (the conversion cv::imread(path, 0) applies before the reference program sees a pixel):
    # derived data, a third of the size, and not a byte copy of the reference's files
    for k in (0, 1, 2, 3):
        g = read_pnm_grey(f"{src}/viff.{k:03d}.ppm").astype(np.uint8)
        with open(f"{dst}/dino_grey_{k:03d}.pgm", "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (g.shape[1], g.shape[0])); f.write(g.tobytes())
    # frames 4..35 complete the 36-view ring of BASELINE configs[4]; same conversion, bzip2 (tests/helpers.py reads them)
    import bz2
    for k in range(4, 36):
        g = read_pnm_grey(f"{src}/viff.{k:03d}.ppm").astype(np.uint8)
        with open(f"{dst}/dino_grey_{k:03d}.pgm.bz2", "wb") as f:
            f.write(bz2.compress(b"P5\n%d %d\n255\n" % (g.shape[1], g.shape[0]) + g.tobytes(), 9))
    imgs = [read_pnm_grey(f"{dst}/dino_grey_{k:03d}.pgm") for k in (0, 1)]
    feats = [O.extract_sift(im, DINO_SIFT["num_octaves"], DINO_SIFT["init_blur"], DINO_SIFT["thresh"], 0.0, False, 32768) for im in imgs]
    n1 = feats[0][1]
    m = O.match_sift(feats[0][0][:n1].copy(), feats[1][0][:feats[1][1]])
    _, _, X0, X1 = O.fill_xu(m, DINO_KINV)
    H = n1 // 8                                                  # the reference's operating point (sfm.cu:95)
    key, counts, Ec = O.ransac_range(X0, X1, 0, H, 1e-6, 0, seed=0x5EED5F3D, want_E=True)      # the library's default seed
    cnt, hyp = O.unpack_key(key)
    out = {"num_pts": np.array([feats[0][1], feats[1][1]]), "stored": np.array([feats[0][2], feats[1][2]]),
           "xpos0": feats[0][0]["xpos"][:n1], "ypos0": feats[0][0]["ypos"][:n1], "scale0": feats[0][0]["scale"][:n1],
           "orientation0": feats[0][0]["orientation"][:n1], "desc0_head": feats[0][0]["data"][:32],
           "match": m["match"], "score": m["score"], "ambiguity": m["ambiguity"],
           "counts": counts, "best": np.array([hyp, cnt]), "E": Ec[hyp]}
    np.savez_compressed(os.path.join(OUT, "dino_oracle.npz"), **out)
    print("dino: features", out["num_pts"], "inliers", cnt, "of", n1, "hypothesis", hyp)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    gen_svd3(); gen_match(); gen_e2e(); gen_dino()
