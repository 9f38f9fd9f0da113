"""CPU: oracle vs the reference compiled in place (oracle/_ref), randomized.  Skipped where the
in-place builds are absent (they need /root/reference at build time)."""
import numpy as np
import pytest

import oracle as O
from helpers import same_bits

need_svd = pytest.mark.skipif(not O.ref_available("libref_svd.so"), reason="oracle/_ref/libref_svd.so not built")
need_match = pytest.mark.skipif(not O.ref_available("libref_match_fma.so"), reason="oracle/_ref/libref_match_fma.so not built")


@need_svd
def test_svd3_random_live():
    R = O.ref_lib("libref_svd.so")
    rng = np.random.default_rng(5)
    fp = lambda a: a.ctypes.data_as(O.f32p)
    for t in range(3000):
        a = (rng.standard_normal(9) * 10 ** rng.uniform(-3, 3)).astype(np.float32)
        if t % 5 == 0:
            a = np.outer(rng.standard_normal(3), rng.standard_normal(3)).astype(np.float32).reshape(9)
        u, s, v = O.svd3(a)
        ru, rs, rv = (np.empty(9, np.float32) for _ in range(3))
        R.ref_svd3(fp(a), fp(ru), fp(rs), fp(rv))
        assert same_bits(u.reshape(9), ru) and same_bits(s.reshape(9), rs) and same_bits(v.reshape(9), rv)
        e = a.copy(); R.ref_normalizeE(fp(e))
        assert same_bits(O.normalizeE(a).reshape(9), e)


@need_match
@pytest.mark.parametrize("n", [64, 300])
def test_match_random_live(n):
    M = O.ref_lib("libref_match_fma.so")
    rng = np.random.default_rng(n)
    d1 = np.abs(rng.standard_normal((n, 128))).astype(np.float32)
    d2 = np.abs(rng.standard_normal((n, 128))).astype(np.float32)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    d1 = np.ascontiguousarray(d1); d2 = np.ascontiguousarray(d2)
    sc = np.zeros(n, np.float32); ix = np.zeros(n, np.int32)
    M.ref_matchC1(n, d1.ctypes.data_as(O.f32p), d2.ctypes.data_as(O.f32p), sc.ctypes.data_as(O.f32p), ix.ctypes.data_as(O.i32p))
    b, s, i = O.match_desc(d1, d2)
    assert np.array_equal(i, ix) and same_bits(b, sc)
