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
