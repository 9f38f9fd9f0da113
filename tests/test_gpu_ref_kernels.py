"""GPU: the REFERENCE'S OWN KERNELS (SfM/kernels.h bodies, CudaSift/matching.cu FindMaxCorr10), compiled
for gfx950 by oracle/ref_build_gpu.sh from the sources where they lie, run on the MI355X and compared
with the oracle and with the product.  This pins the restated kernel bodies to the reference itself.
Skipped when oracle/_ref/libref_kernels.so was not built (needs /root/reference at build time)."""
import ctypes as C

import numpy as np
import pytest

import cuda_sfm_amd as S
import oracle as O
from cuda_sfm_amd_synth import synth
from helpers import same_bits, to_dev

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not O.ref_available("libref_kernels.so"), reason="oracle/_ref/libref_kernels.so not built")]

f32p, i32p = O.f32p, O.i32p


@pytest.fixture(scope="module")
def R(gpu):
    lib = O.ref_lib("libref_kernels.so")
    lib.refk_residual_tail.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_float, i32p]
    lib.refk_vecnorm.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_float, C.c_float]
    return lib


def fp(a):
    return a.ctypes.data_as(f32p)


@pytest.fixture(scope="module")
def scene():
    sc = synth.two_view_scene(1000, seed=17)
    U0, U1, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    return sc, U0, U1, np.ascontiguousarray(X0), np.ascontiguousarray(X1)


def test_copy_point(R, scene):
    sc, U0, U1, _, _ = scene
    n = len(sc["sift"])
    a = np.empty((3, n), np.float32); b = np.empty((3, n), np.float32)
    assert R.refk_copy_point(sc["sift"].ctypes.data_as(C.c_void_p), n, fp(a), fp(b)) == 0
    assert same_bits(a, U0) and same_bits(b, U1)


def test_kron_rows_kernel(R, scene):
    _, _, _, X0, X1 = scene
    n, H = X0.shape[1], 300
    idx = np.array([O.sample8(3, h, n) for h in range(H)], np.int32)
    A = np.empty((H, 72), np.float32)
    assert R.refk_kernels(fp(X0), fp(X1), n, idx.ctypes.data_as(i32p), H, fp(A)) == 0
    ref = np.array([O.build_A(X0, X1, idx[h]).reshape(72) for h in range(H)])
    assert same_bits(A, ref)


def test_row_extraction_and_normalizeE(R, scene):
    _, _, _, X0, X1 = scene
    # testRow_extraction_kernel (sfm.cu:490-502): data[i] = i -> res[b] = 81b+72 .. 81b+80
    Vt = np.arange(81 * 9, dtype=np.float32)
    E = np.empty(9 * 9, np.float32)
    assert R.refk_row_extraction(fp(Vt), 9, fp(E)) == 0
    assert np.array_equal(E.reshape(9, 9), np.array([[81 * b + 72 + k for k in range(9)] for b in range(9)], np.float32))
    # normalizeE kernel vs oracle on real null vectors
    n = X0.shape[1]
    e0 = np.array([O.nullvec9(O.build_A(X0, X1, O.sample8(5, h, n)), 7) for h in range(200)], np.float32)
    got = e0.copy()
    assert R.refk_normalizeE(fp(got), len(got)) == 0
    ref = np.array([O.normalizeE(e).reshape(9) for e in e0])
    assert same_bits(got, ref)


def test_residual_tail_kernels(R, scene):
    """element_wise_div (zero divisor -> 0), element_wise_sum, threshold_count (strict <, NaN never
    counts) as launched at sfm.cu:214-220, fed with the oracle's n^2, da, db."""
    _, _, _, X0, X1 = scene
    n, H, thr = X0.shape[1], 40, np.float32(1e-6)
    n2 = np.empty((H, n), np.float32); da = np.empty_like(n2); db = np.empty_like(n2)
    counts_o = np.empty(H, np.int32)
    for h in range(H):
        Ef = O.hypothesis_E(X0, X1, O.sample8(9, h, n), 7).reshape(9)
        counts_o[h] = O.count_inliers(Ef, X0, X1, thr, want_mask=False)[0]
        n2[h], da[h], db[h] = oracle_terms(Ef, X0, X1)
    counts = np.empty(H, np.int32)
    assert R.refk_residual_tail(fp(n2.copy()), fp(da), fp(n2.copy()), fp(db), n, H, thr, counts.ctypes.data_as(i32p)) == 0
    assert np.array_equal(counts, counts_o)
    # zero divisor and NaN semantics
    nn = np.array([[1.0, 1.0, np.nan, 0.0]], np.float32); dd = np.array([[0.0, 4.0, 1.0, 0.0]], np.float32)
    c = np.empty(1, np.int32)
    assert R.refk_residual_tail(fp(nn.copy()), fp(dd), fp(nn.copy()), fp(dd), 4, 1, np.float32(0.6), c.ctypes.data_as(i32p)) == 0
    assert c[0] == 3          # 0 (zero divisor), 0.5, NaN (not counted), 0


def oracle_terms(E, X0, X1):
    """n^2, da, db exactly as orc_residual forms them (fmaf chains), via float64 emulation of fmaf:
    a float32 fma equals rounding the exact float64 product-sum when the exact result fits in 53 bits,
    which holds for products of two float32 values plus a float32."""
    f = np.float32
    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    E = E.astype(f)
    x1x, x1y, x1z = X0[0], X0[1], X0[2]
    x2x, x2y, x2z = X1[0], X1[1], X1[2]
    a0 = fma(np.full_like(x2y, E[1]), x2y, fma(np.full_like(x2x, E[0]), x2x, (E[2] * x2z).astype(f)))
    a1 = fma(np.full_like(x2y, E[4]), x2y, fma(np.full_like(x2x, E[3]), x2x, (E[5] * x2z).astype(f)))
    a2 = fma(np.full_like(x2y, E[7]), x2y, fma(np.full_like(x2x, E[6]), x2x, (E[8] * x2z).astype(f)))
    b0 = fma(np.full_like(x1y, E[3]), x1y, fma(np.full_like(x1x, E[0]), x1x, (E[6] * x1z).astype(f)))
    b1 = fma(np.full_like(x1y, E[4]), x1y, fma(np.full_like(x1x, E[1]), x1x, (E[7] * x1z).astype(f)))
    nn = fma(x1y, a1, fma(x1x, a0, (a2 * x1z).astype(f)))
    return (nn * nn).astype(f), fma(a1, a1, (a0 * a0).astype(f)), fma(b1, b1, (b0 * b0).astype(f))


def test_vecnorm_literal(R):
    """testVecnorm (sfm.cu:503-510): 1..9 as 3x3, (row=3, col=3, exp=2, pow=1) -> sqrt(66), sqrt(93), sqrt(126)."""
    A = np.arange(1, 10, dtype=np.float32)
    res = np.empty(3, np.float32)
    assert R.refk_vecnorm(fp(A), fp(res), 3, 3, 2.0, 1.0) == 0
    assert np.allclose(res, np.sqrt([66, 93, 126]), rtol=2e-6)       # powf: not correctly rounded
    res2 = np.empty(3, np.float32)
    assert R.refk_vecnorm(fp(A), fp(res2), 3, 3, 2.0, 2.0) == 0      # exp == final_pow: plain sum of squares
    assert np.allclose(res2, [66, 93, 126], rtol=2e-6)


def test_candidate_kernels(R, scene):
    _, _, _, X0, X1 = scene
    n = X0.shape[1]
    for h in range(60):
        E = O.hypothesis_E(X0, X1, O.sample8(2, h, n), 7)
        u, _, v = O.svd3(E)
        if O.det_ref(O.multABt(u, v)) < 0:        # host part of computePosecandidates (sfm.cu:243-245)
            v = -v
        P = np.empty(64, np.float32)
        assert R.refk_candidates(fp(np.ascontiguousarray(u).reshape(9)), fp(np.ascontiguousarray(v).reshape(9)), fp(P)) == 0
        assert same_bits(P, O.pose_candidates(E, O.POSE_REFERENCE).reshape(64))


def test_triangulation_kernels(R, scene):
    _, _, _, X0, X1 = scene
    n = X0.shape[1]
    E = O.hypothesis_E(X0, X1, O.sample8(2, 7, n), 7)
    P = O.pose_candidates(E, O.POSE_REFERENCE)
    I4 = np.eye(4, dtype=np.float32)
    A = np.empty((4, 16), np.float32)
    assert R.refk_tri_A(fp(X0), fp(X1), n, fp(I4.reshape(16)), fp(P.reshape(64)), -1, 1, fp(A)) == 0
    for i in range(4):
        assert same_bits(A[i], O.tri_A(X0[0, 0], X0[1, 0], X1[0, 0], X1[1, 0], I4, P[i]).reshape(16))
    An = np.empty((n, 16), np.float32)
    assert R.refk_tri_A(fp(X0), fp(X1), n, fp(I4.reshape(16)), fp(P.reshape(64)), 2, 0, fp(An)) == 0
    for j in range(0, n, 13):
        assert same_bits(An[j], O.tri_A(X0[0, j], X0[1, j], X1[0, j], X1[1, j], I4, P[2]).reshape(16))
    # normalize_pt_kernal on the oracle's null vectors placed in the last row of V^T (kernels.h:433-450)
    Vt = np.zeros((n, 16), np.float32)
    nv = np.array([O.nullvec4(An[j], 8) for j in range(n)], np.float32)
    nv[5, 3] = 0.0; nv[6, 3] = 7.0
    Vt[:, 12:16] = nv
    pts = np.empty((4, n), np.float32)
    assert R.refk_normalize_pt(fp(Vt), n, fp(pts)) == 0
    ref = np.array([O.normalize_pt(v) for v in nv]).T
    assert same_bits(pts, ref)


@pytest.mark.parametrize("n1,n2", [(512, 512), (1024, 2048)])
def test_findmaxcorr10_vs_product_matcher(R, gpu, n1, n2):
    """The reference's live match kernel on the same GPU: score / match / match_xpos / match_ypos must be
    bit-identical to the product's MFMA matcher (sizes are multiples of 32, so the tail quirk Q1 is not
    exercised); its ambiguity uses an approximate second-best that can only be <= the exact one."""
    torch, dev, ctx = gpu
    d1, _, _ = synth.descriptors(n1, seed=n1)
    d2, _, _ = synth.descriptors(n2, seed=n2 + 1)
    s1 = synth.sift_records(d1, seed=3); s2 = synth.sift_records(d2, seed=4)
    ref = s1.copy()
    assert R.refk_match(ref.ctypes.data_as(C.c_void_p), n1, s2.ctypes.data_as(C.c_void_p), n2) == 0
    t1, t2 = to_dev(torch, dev, s1), to_dev(torch, dev, s2)
    ctx.match(t1, n1, t2, n2)
    torch.cuda.synchronize()
    ours = t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)
    assert np.array_equal(ours["match"], ref["match"])
    for f in ("score", "match_xpos", "match_ypos"):
        assert same_bits(ours[f], ref[f]), f
    assert (ref["ambiguity"] <= ours["ambiguity"] * (1 + 1e-6)).all()
    assert (ref["ambiguity"] == ours["ambiguity"]).mean() > 0.5
    orc = O.match_sift(s1, s2)
    assert np.array_equal(orc["match"], ref["match"]) and same_bits(orc["score"], ref["score"])


@pytest.mark.parametrize("n1,n2", [(512, 512), (1024, 2048), (500, 1037), (96, 40)])
def test_findmaxcorr10_every_field_under_the_quirks(R, gpu, n1, n2):
    """SFM_QUIRK_MATCH_TAIL | SFM_QUIRK_MATCH_AMBIGUITY: the product's sfm_match against the reference's FindMaxCorr10 on the same
    GPU, ALL five fields it writes bit for bit -- `ambiguity` included, whose merge (matching.cu:378-396) ignores seven of the
    eight second-best scores -- on round and ragged sizes; the oracle's restatement agrees; and FindHomography, which gates on
    score and ambiguity (matching.cu:1034-1037), then selects the same matches from either array."""
    torch, dev, ctx = gpu
    d1, _, _ = synth.descriptors(n1, seed=n1 + 7)
    d2, _, _ = synth.descriptors(n2, seed=n2 + 8)
    d2[: min(n1, n2) // 2] = d1[: min(n1, n2) // 2][::-1]              # true partners with a clear margin for half of them
    d2[3] = d2[7 % n2]; d2[32 % n2] = d2[36 % n2]                         # duplicates: runner-up in the same / another row group
    s1 = synth.sift_records(d1, seed=3); s2 = synth.sift_records(d2, seed=4)
    ref = s1.copy()
    assert R.refk_match(ref.ctypes.data_as(C.c_void_p), n1, s2.ctypes.data_as(C.c_void_p), n2) == 0
    qctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
    qctx.set_quirks(S.QUIRK_MATCH_TAIL | S.QUIRK_MATCH_AMBIGUITY)
    t1, t2 = to_dev(torch, dev, s1), to_dev(torch, dev, s2)
    for kern in (S.MATCH_AUTO, S.MATCH_FUSED, S.MATCH_PREFILTER):
        qctx.set_match_kernel(kern)
        t1 = to_dev(torch, dev, s1)
        qctx.match(t1, n1, t2, n2)
        qctx.synchronize()
        ours = t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)
        assert np.array_equal(ours["match"], ref["match"]), kern
        for f in ("score", "match_xpos", "match_ypos", "ambiguity"):
            assert same_bits(ours[f], ref[f]), (f, kern)
    qctx.set_match_kernel(S.MATCH_AUTO)
    # without the ambiguity quirk the product's value is the exact ratio: never below the reference's, and different somewhere
    tctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
    tctx.set_quirks(S.QUIRK_MATCH_TAIL)
    t1x = to_dev(torch, dev, s1)
    tctx.match(t1x, n1, t2, n2)
    tctx.synchronize()
    exact = t1x.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)
    assert (ref["ambiguity"] <= exact["ambiguity"]).all()
    if n2 >= 512:
        assert (ref["ambiguity"] < exact["ambiguity"]).any()
    # the oracle's restatement (its scores are the same fused chains)
    n2s = n2 - n2 % 32
    ob, osec, oi = O.match_second_ref(d1, d2[:n2s]) if n2s else (np.zeros(n1, np.float32), np.zeros(n1, np.float32), -np.ones(n1, np.int32))
    assert same_bits(ob, ref["score"]) and np.array_equal(oi, ref["match"])
    assert same_bits((osec / (ob + np.float32(1e-6))).astype(np.float32), ref["ambiguity"])
    # plain-array entry point: d_second carries the reference's second-best score
    best = torch.empty(n1, dtype=torch.float32, device=dev); sec = torch.empty_like(best); idx = torch.empty(n1, dtype=torch.int32, device=dev)
    if n2s:
        qctx.match_soa(to_dev(torch, dev, d1), n1, 128, to_dev(torch, dev, np.ascontiguousarray(d2[:n2s])), n2s, 128, best, sec, idx)
        qctx.synchronize()
        assert same_bits(sec.cpu().numpy(), osec) and same_bits(best.cpu().numpy(), ob)
    # FindHomography's gate (score > min_score, ambiguity < max_ambiguity) on both arrays: the same subset, hence the same model
    if n1 >= 500:
        gate = lambda a: (a["score"] > 0.85) & (a["ambiguity"] < 0.95)
        assert np.array_equal(gate(ours), gate(ref))
        Hq, nq_ = qctx.find_homography(t1, n1, num_loops=256, seed=5)
        Hr, nr_ = qctx.find_homography(to_dev(torch, dev, ref), n1, num_loops=256, seed=5)
        assert nq_ == nr_ and same_bits(Hq, Hr)


def test_findmaxcorr10_tail_quirk_q1(R, gpu):
    """Quirk Q1 (matching.cu:325): the reference's tile loop never visits the last numPts2 % 32 points of the
    second set.  The product visits all of them; restricted to the points the reference does visit it
    reproduces the reference bit for bit, and the full call differs exactly where a skipped point wins."""
    torch, dev, ctx = gpu
    n1, n2 = 512, 1024 + 13
    d1, _, _ = synth.descriptors(n1, seed=21)
    d2, _, _ = synth.descriptors(n2, seed=22)
    d2[-13:] = d1[:13]                                        # the skipped tail holds the true partners of points 0..12
    s1 = synth.sift_records(d1, seed=3); s2 = synth.sift_records(d2, seed=4)
    ref = s1.copy()
    assert R.refk_match(ref.ctypes.data_as(C.c_void_p), n1, s2.ctypes.data_as(C.c_void_p), n2) == 0
    t2 = to_dev(torch, dev, s2)
    t1 = to_dev(torch, dev, s1)
    ctx.match(t1, n1, t2, n2 - n2 % 32)                      # what the reference actually searches
    torch.cuda.synchronize()
    cut = t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)
    assert np.array_equal(cut["match"], ref["match"]) and same_bits(cut["score"], ref["score"])
    assert (ref["match"] < n2 - 13).all()
    t1 = to_dev(torch, dev, s1)
    ctx.match(t1, n1, t2, n2)                                # the product's answer
    torch.cuda.synchronize()
    full = t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)
    assert np.array_equal(full["match"][:13], np.arange(n2 - 13, n2)) and (full["score"][:13] > 0.999).all()
    rest = full["match"] < n2 - 13                          # wherever no skipped point wins the two agree
    assert rest[13:].mean() > 0.9 and np.array_equal(full["match"][rest], ref["match"][rest]) and same_bits(full["score"][rest], ref["score"][rest])
    # SFM_QUIRK_MATCH_TAIL: the product emulates the skipped tail itself, for A/B runs against the reference
    qctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
    qctx.set_quirks(S.QUIRK_MATCH_TAIL)
    t1 = to_dev(torch, dev, s1)
    qctx.match(t1, n1, t2, n2)
    torch.cuda.synchronize()
    quirk = t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)
    assert np.array_equal(quirk["match"], ref["match"]) and same_bits(quirk["score"], ref["score"])
    for f in ("match_xpos", "match_ypos"):
        assert same_bits(quirk[f], ref[f])
    t1 = to_dev(torch, dev, s1)
    qctx.match(t1, n1, t2, 20)                               # fewer than 32 points: nothing is searched
    torch.cuda.synchronize()
    none = t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)
    assert (none["match"] == -1).all() and not none["score"].any()
    qctx.set_quirks(0)
    with pytest.raises(S.SfmError):
        qctx.set_quirks(8)
