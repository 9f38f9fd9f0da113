"""CPU: the product's device arithmetic (cuda-sfm_amd/csrc/device_math.hpp), compiled as HIP *host*
code by tests/hostcheck, against the oracle -- bit for bit.  This is how kernel math is validated in
a container without a GPU; tests/hostcheck is a test harness, not a fallback path."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
from cuda_sfm_amd_synth import synth
from helpers import same_bits

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck", "libhostcheck.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="tests/hostcheck not built (make hostcheck)")

f32p, i32p = O.f32p, O.i32p


@pytest.fixture(scope="module")
def H():
    h = C.CDLL(LIB)
    h.hc_residual.restype = C.c_float
    h.hc_residual.argtypes = [f32p] + [C.c_float] * 6
    h.hc_inlier_filter.restype = C.c_int
    h.hc_inlier_filter.argtypes = [f32p, C.c_float] + [C.c_float] * 6
    h.hc_pack_key.restype = C.c_uint64
    h.hc_triangulate_point.argtypes = [C.c_float] * 4 + [f32p, C.c_int, f32p]
    return h


@pytest.fixture(scope="module")
def scene():
    sc = synth.two_view_scene(2048, seed=31)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    return np.ascontiguousarray(X0), np.ascontiguousarray(X1)


def fp(a):
    return a.ctypes.data_as(f32p)


def test_sampler(H):
    """The product's slot-by-slot sampler against the oracle's plain loop: same candidate sequence, same ids -- including sets
    so small that nearly every candidate repeats (n = 8 .. 12: dozens of redraws per sample) and sets of fewer than 8 points,
    where 256 candidates cannot yield 8 ids and the sequential fallback fills the rest."""
    b = np.empty(8, np.int32)
    for n in (1, 2, 5, 7, 8, 9, 10, 12, 16, 33, 100, 2155, 4096, 1 << 20):
        hyps = list(range(300 if n <= 100 else 60)) + [65535, 2 ** 31 + 5, 2 ** 32 - 1]
        for seed in (0x5EED5F3D, 0, 12345):
            for h in hyps:
                a = O.sample8(seed, h, n)
                H.hc_sample8(C.c_uint32(seed), C.c_uint32(h), n, b.ctypes.data_as(i32p))
                assert np.array_equal(a, b), (n, seed, h, a, b)
                if n >= 8: assert len(set(b.tolist())) == 8 and b.min() >= 0 and b.max() < n


def test_hypothesis_E_bit_exact(H, scene):
    X0, X1 = scene
    n = X0.shape[1]
    for sweeps in (0, 1, 4, 7):                     # 0 = Householder solver
        for h in range(200):
            idx = O.sample8(5, h, n)
            E = np.empty(9, np.float32)
            H.hc_hypothesis_E(fp(X0), fp(X1), n, idx.ctypes.data_as(i32p), sweeps, fp(E))
            assert same_bits(E, O.hypothesis_E(X0, X1, idx, sweeps).reshape(9)), (sweeps, h)


def test_packed_two_hypotheses_per_lane(H, scene):
    """The v2f instantiation (two hypotheses per lane, v_pk_* math on the GPU) is bit-identical per
    hypothesis to the scalar one and to the oracle, including mixed degenerate / regular pairs."""
    X0, X1 = scene
    n = X0.shape[1]
    tuples = [O.sample8(6, h, n) for h in range(60)] + [np.array([0] * 8, np.int32), np.array([3, 3, 4, 4, 5, 5, 6, 6], np.int32)]
    for a in range(0, len(tuples) - 1):
        ia, ib = tuples[a], tuples[(a * 7 + 3) % len(tuples)]
        EA = np.empty(9, np.float32); EB = np.empty(9, np.float32)
        H.hc_hypothesis_E_pair(fp(X0), fp(X1), n, ia.ctypes.data_as(i32p), ib.ctypes.data_as(i32p), 7, fp(EA), fp(EB))
        assert same_bits(EA, O.hypothesis_E(X0, X1, ia, 7).reshape(9)), a
        assert same_bits(EB, O.hypothesis_E(X0, X1, ib, 7).reshape(9)), a


def test_degenerate_tuples(H, scene):
    X0, X1 = scene
    n = X0.shape[1]
    for idx in ([0] * 8, [0, 0, 1, 1, 2, 2, 3, 3], [5, 6, 7, 8, 9, 10, 11, 12]):
        idx = np.array(idx, np.int32)
        E = np.empty(9, np.float32)
        H.hc_hypothesis_E(fp(X0), fp(X1), n, idx.ctypes.data_as(i32p), 7, fp(E))
        assert same_bits(E, O.hypothesis_E(X0, X1, idx, 7).reshape(9))


def test_residual_and_filter_are_exact(H, scene):
    """The division-free filter may answer 'undecided' but never disagrees with residual() < thr."""
    X0, X1 = scene
    n = X0.shape[1]
    thr = np.float32(1e-6)
    und = 0
    for h in range(40):
        E = O.hypothesis_E(X0, X1, O.sample8(8, h, n), 7).reshape(9)
        _, mask = O.count_inliers(E, X0, X1, thr)
        for j in range(0, n, 3):
            args = [float(v) for v in (*X0[:, j], *X1[:, j])]
            r = H.hc_residual(fp(E), *args)
            assert same_bits(np.array([r], np.float32), np.array([O.residual(E, X0[:, j], X1[:, j])], np.float32))
            f = H.hc_inlier_filter(fp(E), thr, *args)
            und += f < 0
            assert f < 0 or f == mask[j]
    assert und < 20


def test_filter_adversarial_thresholds(H, scene):
    """Put the threshold right at (and 1 ulp around) each point's exact residual: the filter must
    say 'undecided' or agree; also zero / NaN / huge operands."""
    X0, X1 = scene
    E = O.hypothesis_E(X0, X1, O.sample8(8, 3, X0.shape[1]), 7).reshape(9)
    for j in range(0, 600):
        args = [float(v) for v in (*X0[:, j], *X1[:, j])]
        r = np.float32(O.residual(E, X0[:, j], X1[:, j]))
        if not np.isfinite(r) or r <= 0:
            continue
        for thr in (r, np.nextafter(r, np.float32(0)), np.nextafter(r, np.float32(1)), np.float32(r * (1 + 3e-6)), np.float32(r * (1 - 3e-6))):
            f = H.hc_inlier_filter(fp(E), thr, *args)
            assert f < 0 or f == int(r < thr), (j, r, thr, f)
    Z = np.zeros(9, np.float32)
    assert H.hc_inlier_filter(fp(Z), np.float32(1e-6), 1, 2, 1, 3, 4, 1) < 0          # zero divisors -> exact path
    nanE = np.full(9, np.nan, np.float32)
    assert H.hc_inlier_filter(fp(nanE), np.float32(1e-6), 1, 2, 1, 3, 4, 1) < 0
    big = (E * np.float32(1e20)).astype(np.float32)
    assert H.hc_inlier_filter(fp(big), np.float32(1e-6), 1, 2, 1, 3, 4, 1) < 0         # da*db overflows the guard
    for thr in (0.0, 1e-30, 1e30, np.inf):
        assert H.hc_inlier_filter(fp(E), np.float32(thr), *[float(v) for v in (*X0[:, 0], *X1[:, 0])]) < 0


def test_nullvec4_skipped_rotations(H):
    """Column pairs that are exactly orthogonal are left alone by the 4x4 Jacobi (`if (ga == 0) continue;` in the oracle, selects on
    the results in the product): matrices with orthogonal, zero, signed-zero, duplicate and NaN / infinite columns -- every bit of
    the null vector equal, signs of zeros included."""
    rng = np.random.default_rng(5)
    cases = [np.eye(4), np.diag([3.0, -2.0, 0.5, 0.0]), np.eye(4)[[1, 0, 3, 2]], np.zeros((4, 4)), -np.zeros((4, 4)),
             np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1], [0, 0, 1, 1.0]]),
             np.array([[1, 1, 0, 0], [1, -1, 0, 0], [0, 0, 2, 0], [0, 0, 0, -0.0]]),
             np.array([[1, 2, 3, 4], [0, 0, 0, 0], [5, 6, 7, 8], [0, -0.0, 0, 0.0]])]
    for _ in range(40):                                      # random matrices with some columns zeroed / some blocks decoupled
        A = rng.standard_normal((4, 4))
        if rng.random() < 0.5: A[:, rng.integers(0, 4)] = 0.0
        if rng.random() < 0.5: A[:2, 2:] = 0.0; A[2:, :2] = 0.0
        if rng.random() < 0.3: A[rng.integers(0, 4), :] = -0.0
        cases.append(A)
    nanA = rng.standard_normal((4, 4)); nanA[1, 2] = np.nan; cases.append(nanA)
    infA = rng.standard_normal((4, 4)); infA[3, 0] = np.inf; cases.append(infA)
    for A in cases:
        A = np.ascontiguousarray(A, np.float32).reshape(16)
        for sweeps in (1, 8):
            v = np.empty(4, np.float32); H.hc_nullvec4(fp(A), sweeps, fp(v))
            assert same_bits(v, O.nullvec4(A, sweeps)), (A, sweeps)


def test_pose_and_4x4(H, scene):
    X0, X1 = scene
    rng = np.random.default_rng(0)
    for h in range(100):
        E = O.hypothesis_E(X0, X1, O.sample8(2, h, X0.shape[1]), 7)
        for mode in (0, 1):
            P = np.empty(64, np.float32)
            H.hc_pose_candidates(fp(E.reshape(9)), mode, fp(P))
            assert same_bits(P, O.pose_candidates(E, mode).reshape(64))
        A = rng.standard_normal(16).astype(np.float32)
        v = np.empty(4, np.float32); H.hc_nullvec4(fp(A), 8, fp(v))
        assert same_bits(v, O.nullvec4(A, 8))
        inv = np.zeros(16, np.float32)
        assert H.hc_inv4(fp(A), fp(inv)) == 1 and same_bits(inv, O.inv4(A)[1].reshape(16))
        Pm = O.pose_candidates(E, 0)[h % 4].reshape(16)
        out = np.empty(4, np.float32)
        H.hc_triangulate_point(float(X0[0, h]), float(X0[1, h]), float(X1[0, h]), float(X1[1, h]), fp(Pm), 8, fp(out))
        ref = O.triangulate(X0[:, h:h + 1], X1[:, h:h + 1], Pm, 8)[:, 0]
        assert same_bits(out, ref)
    assert H.hc_pack_key(C.c_uint32(7), C.c_uint32(3)) == O.pack_key(7, 3)


def test_binary32_reciprocal_sqrt_equals_the_double_promoted_form(H):
    """svd.h evaluates 1.0/sqrtf(x) and x*1.0/sqrtf(x) in double and rounds to float (svd.h:33-36, 129, 250).  The
    product code uses plain binary32 division; a double rounding cannot change the result (device_math.hpp).  Checked
    here on 4 million floats of every magnitude plus the special values."""
    rng = np.random.default_rng(12)
    bits = rng.integers(0, 0x7F800000, 4_000_000, dtype=np.uint32)            # every positive finite float is equally likely
    x = np.concatenate([bits.view(np.float32), np.array([0.0, 1.0, 4.0, 2.0, 3.0, 1e-45, 1.17549435e-38, 3.4028235e38, np.inf, -1.0, np.nan], np.float32)])
    rs = np.empty_like(x); acs = np.empty_like(x)
    H.hc_rsqrt_forms(fp(x), fp(rs), fp(acs), len(x))
    with np.errstate(all="ignore"):
        s = np.sqrt(x).astype(np.float64)                                      # sqrtf: correctly rounded
        want_rs = (1.0 / s).astype(np.float32)
        want_as = ((x.astype(np.float64) * 1.0) / s).astype(np.float32)
    assert same_bits(rs, want_rs)
    assert same_bits(acs, want_as)
