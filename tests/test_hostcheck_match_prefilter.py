"""CPU: the error bound of the pre-filter matcher (cuda-sfm_amd/csrc/match_prefilter_math.hpp, compiled as HIP host code by
tests/hostcheck).  match_pf_eps(|a|, |b|) must cover |fp16 matrix-core score - exact fp32 chain| for every pair of rows:
the fp16 copies are taken from the product's own conversion, their dot product is evaluated in float64 and pushed by the
whole fp32-accumulation budget of the eight MFMAs; the exact chain is the oracle's.  (GPU twin: tests/test_gpu_match_prefilter.py,
results bit for bit.)"""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import oracle as O

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck", "libhostcheck.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="tests/hostcheck not built (make hostcheck)")
f32p = O.f32p
ACC = 8 * 2.0 ** -20            # budget for the eight chained MFMAs; measured on MI355X: 1.2 * 2^-24 per instruction


@pytest.fixture(scope="module")
def H():
    h = C.CDLL(LIB)
    h.hc_match_pf_eps.restype = C.c_float
    h.hc_match_pf_eps.argtypes = [C.c_float, C.c_float]
    h.hc_match_pf_norm_up.restype = C.c_float
    h.hc_match_pf_norm_up.argtypes = [f32p]
    h.hc_match_pf_half.argtypes = [f32p, f32p, C.c_int]
    return h


def half_copy(H, x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    H.hc_match_pf_half(x.ctypes.data_as(f32p), out.ctypes.data_as(f32p), x.size)
    return out


def chain(a, b):
    """fmaf(a[127], b[127], ... fmaf(a[0], b[0], 0)): the product is exact in float64, one rounding to float32 per step
    (a double rounding can differ from fmaf by one ulp in a halfway case: far inside the bound's slack)"""
    s = np.float32(0.0)
    for d in range(128):
        s = np.float32(np.float64(a[d]) * np.float64(b[d]) + np.float64(s))
    return s


def rows(rng, kind, n):
    if kind == "sift":                       # unit norm, non-negative, clipped at 0.2 like CudaSift's descriptors
        x = np.abs(rng.standard_normal((n, 128))) ** 3
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        x = np.minimum(x, 0.2); x /= np.linalg.norm(x, axis=1, keepdims=True)
    elif kind == "signed":
        x = rng.standard_normal((n, 128)) * 10.0 ** rng.uniform(-3, 1, (n, 1))
    elif kind == "tiny":                     # mostly below the fp16 floor of the scaled copy (2^-14 / 2^8 = 2.4e-7)
        x = rng.standard_normal((n, 128)) * 10.0 ** rng.uniform(-9, -5, (n, 128))
    elif kind == "spiky":                    # a few large entries (up to the 255 limit) among small ones
        x = rng.standard_normal((n, 128)) * 1e-3
        for r in range(n):
            x[r, rng.integers(0, 128, 3)] = rng.uniform(-255, 255, 3)
    else:                                    # halfway cases of the fp16 rounding: k + 1/2 ulp patterns
        k = rng.integers(1024, 2048, (n, 128)).astype(np.float64)
        x = (k + 0.5) * 2.0 ** rng.integers(-18, -6, (n, 128)) / 256.0 * rng.choice([-1.0, 1.0], (n, 128))
    return x.astype(np.float32)


@pytest.mark.parametrize("kind_a,kind_b", [("sift", "sift"), ("signed", "signed"), ("tiny", "sift"), ("tiny", "tiny"),
                                           ("spiky", "signed"), ("spiky", "spiky"), ("half", "half"), ("half", "sift")])
def test_eps_covers_the_fp16_score(H, kind_a, kind_b):
    rng = np.random.default_rng(zlib.crc32((kind_a + "|" + kind_b).encode()))
    A, B = rows(rng, kind_a, 40), rows(rng, kind_b, 40)
    hA, hB = half_copy(H, A).astype(np.float64), half_copy(H, B).astype(np.float64)
    worst = 0.0
    for i in range(A.shape[0]):
        na = H.hc_match_pf_norm_up(A[i].ctypes.data_as(f32p))
        assert na >= np.linalg.norm(A[i].astype(np.float64))
        for j in range(B.shape[0]):
            nb = H.hc_match_pf_norm_up(B[j].ctypes.data_as(f32p))
            eps = float(H.hc_match_pf_eps(na, nb))
            approx = float(np.dot(hA[i], hB[j]))
            budget = ACC * float(np.dot(np.abs(hA[i]), np.abs(hB[j])))
            err = abs(approx - float(chain(A[i], B[j]))) + budget
            assert err <= eps, (kind_a, kind_b, i, j, err, eps)
            worst = max(worst, err / eps)
    assert worst > 1e-4            # the bound is not vacuous on these inputs


def test_eps_monotone_and_infinite(H):
    assert H.hc_match_pf_eps(1.0, 1.0) < H.hc_match_pf_eps(1.0, 2.0) < H.hc_match_pf_eps(3.0, 2.0)
    assert np.isinf(H.hc_match_pf_eps(np.inf, 1.0)) and np.isinf(H.hc_match_pf_eps(1.0, np.inf))
    assert not (H.hc_match_pf_eps(np.inf, 0.0) < np.inf)         # inf * 0: NaN, which the kernel treats as "no bound"


def chain_all(A, B):
    """the exact chain of every (row of A, row of B) pair at once (float64 product, one rounding to float32 per step)"""
    s = np.zeros((A.shape[0], B.shape[0]), np.float32)
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    for d in range(128):
        s = (A64[:, d:d + 1] * B64[None, :, d] + s.astype(np.float64)).astype(np.float32)
    return s


def reference_fold(scores):
    """FindMaxCorr10's rule over one query's scores in row order (matching.cu:352-361): (best, second, index)"""
    best, second, idx = np.float32(0.0), np.float32(0.0), -1
    for p, s in enumerate(scores):
        if s > best:
            second, best, idx = best, s, p
        elif s > second:
            second = s
    return best, second, idx


@pytest.mark.parametrize("kind_q,kind_db,seed", [("sift", "sift", 1), ("sift", "sift", 2), ("signed", "signed", 3), ("spiky", "sift", 4),
                                                 ("half", "half", 5), ("tiny", "sift", 6)])
def test_fused_running_threshold_lists_every_row_that_matters(H, kind_q, kind_db, seed):
    """The listing rule of match_fused.hip on the CPU: stages of 128 rows, the maxima of the eight sixteen-row sets a lane pair
    owns per stage, A2 = the second largest of all set maxima so far, tau = A2 - 2 eps (eps with the largest row norm so far),
    listed = rows of the stage with approximate score >= tau.  The approximate scores are the fp16 copies' dot products pushed
    ADVERSARIALLY by the whole accumulation budget (down for the rows that matter, up for the others).  Folding the listed
    rows' exact scores with the reference's rule must give the reference's (best, second, index) over ALL rows."""
    rng = np.random.default_rng(seed)
    nq, ndb = 24, 700                                     # 5.5 stages: a short last one included
    Q, D = rows(rng, kind_q, nq), rows(rng, kind_db, ndb)
    if kind_q == "sift":                                  # near-duplicates of some rows, as matched views have
        for i in range(0, nq, 2):
            Q[i] = D[rng.integers(0, ndb)] + 1e-3 * rng.standard_normal(128).astype(np.float32)
    D[333] = D[40]; D[650] = D[40]                        # exact duplicates in different stages
    exact = chain_all(Q, D)
    hQ, hD = half_copy(H, Q).astype(np.float64), half_copy(H, D).astype(np.float64)
    approx = hQ @ hD.T
    budget = ACC * (np.abs(hQ) @ np.abs(hD).T)
    qn = np.array([H.hc_match_pf_norm_up(Q[i].ctypes.data_as(f32p)) for i in range(nq)], np.float32)
    dn = np.array([H.hc_match_pf_norm_up(D[j].ctypes.data_as(f32p)) for j in range(ndb)], np.float32)
    listed_total = 0
    for i in range(nq):
        want = reference_fold(exact[i])
        # adversary: rows that take part in the reference's result look as small as the budget allows, all others as large
        matters = exact[i] >= want[1]
        a = np.where(matters, approx[i] - budget[i], approx[i] + budget[i])
        A1 = A2 = 0.0
        bmax = np.float32(0.0)
        listed = []
        for s0 in range(0, ndb, 128):
            rows_s = np.arange(s0, min(ndb, s0 + 128))
            bmax = max(bmax, dn[rows_s].max())
            padded = np.zeros(128); padded[:rows_s.size] = a[rows_s]            # rows beyond the end are staged as zeros
            for rt in range(4):
                for half in range(2):
                    members = [32 * rt + (r & 3) + 8 * (r >> 2) + 4 * half for r in range(16)]
                    m = max(0.0, padded[members].max())
                    A2 = max(min(A1, m), A2)
                    A1 = max(A1, m)
            eps = float(H.hc_match_pf_eps(qn[i], bmax))
            tau = A2 - 2.0 * eps
            tau = tau - abs(tau) * 1.2e-7
            listed += [int(r) for r in rows_s if not (a[r] < tau)]
        listed_total += len(listed)
        got = reference_fold(exact[i][listed])
        got = (got[0], got[1], listed[got[2]] if got[2] >= 0 else -1)
        assert (got[0], got[1], got[2]) == (want[0], want[1], want[2]), (kind_q, kind_db, i, got, want, len(listed))
    assert listed_total < nq * ndb // 2 or kind_q in ("tiny", "half")          # the rule does filter on ordinary data
