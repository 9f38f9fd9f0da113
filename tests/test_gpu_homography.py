"""GPU: FindHomography (reference CudaSift/matching.cu:1000-1087).
 - the oracle (orc_homography4 / orc_homography_count) against the REFERENCE'S OWN ComputeHomographies /
   TestHomographies / InvertMatrix<8> compiled for gfx950 from matching.cu in place (oracle/_ref) -- bit-exact;
 - the product (sfm_find_homography through the C ABI) against the oracle -- bit-exact for every hypothesis,
   every support count and the winner, including ragged point counts the reference cannot test exactly."""
import ctypes as C

import numpy as np
import pytest

import cuda_sfm_amd as S
import oracle as O
from cuda_sfm_amd_synth import synth
from helpers import same_bits, to_dev

pytestmark = pytest.mark.gpu


def coords(sift, ld=None):
    n = len(sift)
    ld = ld or n
    c = np.full((4, ld), np.nan, np.float32)
    for r, k in enumerate(("xpos", "ypos", "match_xpos", "match_ypos")):
        c[r, :n] = sift[k]
    return c


def sample(n, L, seed):
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(np.array([rng.choice(n, 4, replace=False) for _ in range(L)], np.int32).T)


def oracle_all(c, n, pts, thresh):
    L = pts.shape[1]
    homo = np.array([O.homography4(c, pts[:, l]) for l in range(L)], np.float32)
    cnt = np.array([O.homography_count(homo[l], c, n, np.float32(thresh) * np.float32(thresh)) for l in range(L)], np.int32)
    return np.ascontiguousarray(homo.T), cnt


@pytest.mark.skipif(not O.ref_available("libref_kernels.so"), reason="oracle/_ref/libref_kernels.so not built")
@pytest.mark.parametrize("n,L,thresh", [(1024, 256, 5.0), (48, 32, 2.0), (4096, 64, 1.0)])
def test_oracle_matches_reference_kernels(gpu, n, L, thresh):
    sc = synth.homography_scene(n, seed=3 + n)
    c = coords(sc["sift"])
    pts = sample(n, L, 11)
    R = O.ref_lib("libref_kernels.so")
    R.refk_homography.argtypes = [O.f32p, C.c_int, O.i32p, C.c_int, C.c_float, O.f32p, O.i32p]
    homo = np.empty((8, L), np.float32); cnt = np.empty(L, np.int32)
    assert R.refk_homography(c.ctypes.data_as(O.f32p), n, pts.ctypes.data_as(O.i32p), L, thresh,
                             homo.ctypes.data_as(O.f32p), cnt.ctypes.data_as(O.i32p)) == 0
    oh, oc = oracle_all(c, n, pts, thresh)
    assert same_bits(homo, oh)
    assert np.array_equal(cnt, oc)
    if thresh >= 5.0:
        assert cnt.max() > 0.4 * n          # the scene has a dominant plane


@pytest.mark.parametrize("n,L,thresh", [(1024, 256, 5.0), (1000, 100, 3.0), (77, 16, 5.0), (5000, 1000, 5.0), (8, 16, 5.0)])
def test_product_matches_oracle(gpu, n, L, thresh):
    sc = synth.homography_scene(n, seed=5 + n)
    torch, dev, ctx = gpu
    d = to_dev(torch, dev, sc["sift"])
    Lup = (L + 15) // 16 * 16
    pts = sample(n, Lup, 13)
    H, nm, cnt, homo = ctx.find_homography(d, n, num_loops=L, thresh=thresh, pts=pts, want_all=True)
    c = coords(sc["sift"])
    oh, oc = oracle_all(c, n, pts, thresh)
    assert same_bits(homo, oh)
    assert np.array_equal(cnt, oc)
    best = int(np.argmax(oc))               # first maximum (matching.cu:1066-1070)
    assert nm == oc[best]
    assert same_bits(H.reshape(9)[:8], oh[:, best]) and H[2, 2] == 1.0


def test_seeded_sampler_gate_and_recovery(gpu):
    n = 3000
    sc = synth.homography_scene(n, seed=21)
    s = sc["sift"]
    torch, dev, ctx = gpu
    d = to_dev(torch, dev, s)
    H, nm, cnt, homo = ctx.find_homography(d, n, num_loops=1000, seed=7, want_all=True)
    H2, nm2 = ctx.find_homography(d, n, num_loops=1000, seed=7)
    assert same_bits(H, H2) and nm == nm2   # deterministic
    H3, nm3 = ctx.find_homography(d, n, num_loops=1000, seed=8)
    assert not same_bits(H, H3)
    # recovers the plane: support ~ inlier fraction, H close to truth in transfer error
    assert nm > 0.9 * (~sc["outlier"]).sum()
    x = np.array([[100, 100, 1], [1800, 200, 1], [900, 1000, 1]], np.float64).T
    a = sc["H"].astype(np.float64) @ x; b = H.astype(np.float64) @ x
    assert np.abs(a[:2] / a[2] - b[:2] / b[2]).max() < 5.0      # within the 5 px support threshold
    # every hypothesis is an exact fit of four GATED points: recompute the sample's support through the oracle
    c = coords(s)
    valid = (s["score"] > np.float32(0.85)) & (s["ambiguity"] < np.float32(0.95))
    assert 0.3 * n < valid.sum() < 0.95 * n
    thr2 = np.float32(25.0)
    for l in (0, 1, 500, 999):
        assert cnt[l] == O.homography_count(homo[:, l], c, n, thr2)
        # the four generating points have (near) zero error -> they are inliers and gated
        hl = np.append(homo[:, l], 1).reshape(3, 3).astype(np.float64)
        p = hl @ np.vstack([c[0, :n], c[1, :n], np.ones(n)])
        err = np.hypot(p[0] / p[2] - c[2, :n], p[1] / p[2] - c[3, :n])
        four = np.argsort(err)[:4]                          # binary32 LU: the fit is exact to a fraction of a pixel
        assert err[four].max() < 0.5 and valid[four].all()


def test_degenerate_inputs(gpu):
    torch, dev, ctx = gpu
    sc = synth.homography_scene(64, seed=2)
    d = to_dev(torch, dev, sc["sift"])
    H, nm = ctx.find_homography(d, 7)                     # < 8 points -> identity, 0 (matching.cu:1010-1014)
    assert np.array_equal(H, np.eye(3, dtype=np.float32)) and nm == 0
    s = sc["sift"].copy(); s["score"] = 0.1               # nothing passes the gate (matching.cu:1037)
    H, nm = ctx.find_homography(to_dev(torch, dev, s), 64)
    assert np.array_equal(H, np.eye(3, dtype=np.float32)) and nm == 0
    with pytest.raises(S.SfmError):
        ctx.find_homography(d, 64, num_loops=16, pts=np.full((4, 16), 64, np.int32))


@pytest.mark.parametrize("n,loops,ms,ma,thr,seed", [(3000, 1000, 0.85, 0.95, 5.0, 7), (2048, 10000, 0.0, 0.80, 5.0, 0),
                                                    (100, 17, 0.85, 0.95, 2.0, 99), (9, 16, 0.0, 1.1, 5.0, 1)])
def test_seeded_end_to_end_matches_oracle(gpu, n, loops, ms, ma, thr, seed):
    """whole FindHomography (gate, keyed sample, DLT, support, first maximum) == oracle, bit for bit"""
    torch, dev, ctx = gpu
    s = synth.homography_scene(n, seed=31 + n)["sift"]
    H, nm, cnt, homo = ctx.find_homography(to_dev(torch, dev, s), n, loops, ms, ma, thr, seed, want_all=True)
    oH, onm, ocnt, ohomo = O.find_homography(s, loops, ms, ma, thr, seed, want_all=True)
    assert same_bits(homo, ohomo) and np.array_equal(cnt, ocnt)
    assert nm == onm and same_bits(H, oH)
