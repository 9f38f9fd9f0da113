"""CPU: the matrix-core pre-filter's operands and decision rules (cuda-sfm_amd/csrc/prefilter_math.hpp, compiled as HIP host
code by tests/hostcheck) against the oracle.  A rule may only REJECT pairs the exact test would not count; here the fp16
contractions are evaluated in float64 and then pushed by the full accumulation-error budget in every direction that
favours a rejection -- a single rejected oracle inlier fails the test.  All rules run: "pack" (round 6, the product: the band
rule -- a per-hypothesis constant folded into the coefficient scaling -- scanned with the six-bit conversion: rejected <=> |nt| >=
1.875), "band" (round 5: the same rule scanned with v_alignbit_b32, the test is bit 30 of the accumulator) and "G" (rounds 2-4: a
per-pair threshold from a second contraction); the last two are kept in the lab-bench library.  (The GPU twin is tests/test_gpu_prefilter.py:
counts bit for bit.)"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
from cuda_sfm_amd_synth import synth

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck", "libhostcheck.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="tests/hostcheck not built (make hostcheck)")
f32p = O.f32p
ACC = 8 * 2.0 ** -24            # budget per contraction; measured on MI355X: 1.2 * 2^-24 per instruction (profiles/probes/mfma_f16_probe.hip)


@pytest.fixture(scope="module")
def H():
    h = C.CDLL(LIB)
    h.hc_pf_scales.argtypes = [C.c_float, C.POINTER(C.c_int), f32p, f32p, f32p]
    h.hc_pf_hyp_slots.restype = C.c_float
    h.hc_pf_hyp_slots.argtypes = [f32p, C.c_float, C.c_float, C.c_int, f32p, f32p]
    h.hc_pf_point_slots.argtypes = [C.c_float] * 4 + [C.c_int, f32p, f32p]
    h.hc_pf_reject.argtypes = [C.c_float] * 2
    h.hc_pf_zero_divisor_cells.argtypes = [f32p, C.c_float, C.POINTER(C.c_int), f32p]
    h.hc_pf_point_cell.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
    h.hc_pf_cell_key.restype = C.c_uint32
    h.hc_pf_cell_key.argtypes = [C.c_int, C.c_int]
    h.hc_pf_zero_divisor_any.argtypes = [f32p, f32p, f32p, C.c_int]
    h.hc_pf_band_sigma.restype = C.c_float
    h.hc_pf_band_sigma.argtypes = [f32p, C.c_float, C.c_float, f32p, C.c_int]
    h.hc_pf_band_hyp_slots.argtypes = [f32p, C.c_float, f32p]
    h.hc_pf_band_reject.argtypes = [C.c_float]
    h.hc_pf_band_sigma_top.restype = C.c_float
    h.hc_pf_band_sigma_top.argtypes = [f32p, C.c_float, C.c_float, f32p, C.c_int, C.c_float]
    h.hc_pf_band_top.restype = C.c_float
    h.hc_pf_band_top.argtypes = [C.c_int]
    h.hc_pf_band_pack_reject.argtypes = [C.c_float]
    h.hc_pf_pack_code.restype = C.c_uint32
    h.hc_pf_pack_code.argtypes = [C.c_int]
    h.hc_pf_pack_field.restype = C.c_uint32
    h.hc_pf_pack_field.argtypes = [C.c_int]
    h.hc_pf_tile_sigma.restype = C.c_float
    h.hc_pf_tile_sigma.argtypes = [f32p, C.c_float, C.c_float, f32p, C.c_int]
    h.hc_pf_morton_key.restype = C.c_uint32
    h.hc_pf_morton_key.argtypes = [C.c_float] * 6
    h.hc_pf_transposed.argtypes = [f32p, f32p]
    h.hc_pf_cell_key_side.restype = C.c_uint32
    h.hc_pf_cell_key_side.argtypes = [C.c_int, C.c_int, C.c_int]
    h.hc_pf_order_bits.restype = C.c_uint32
    h.hc_pf_order_bits.argtypes = [C.c_float]
    h.hc_pf_box_from_words.argtypes = [C.POINTER(C.c_uint64), C.c_float, f32p]
    h.hc_pf_box_from_bound.argtypes = [C.POINTER(C.c_uint64), C.c_float, f32p]
    return h


RULES = ["pack", "tile", "band", "G"]
BAND_RULES = ("pack", "tile", "band")
HW_ULPS = 1.0 - 8 * 2.0 ** -23      # "tile": the device's 1-ulp reciprocal / square root can move sigma by a few ulp; the rule must hold for the LARGEST sigma the device can derive


def band_sigma(H, e, thr, B, box, b_safe, rule):
    """sigma of a hypothesis under either scan of the band rule (top = 1.998 / 1.873); "tile": the per-tile variant's cheap form, pushed
    UP by the hardware operations' error so that the check covers whatever the device computes."""
    if rule == "tile":
        s = float(H.hc_pf_tile_sigma(fp(np.ascontiguousarray(e, np.float32)), thr, B, fp(box), int(b_safe)))
        exact = float(H.hc_pf_band_sigma_top(fp(np.ascontiguousarray(e, np.float32)), thr, B, fp(box), int(b_safe), H.hc_pf_band_top(1)))
        assert s <= exact * (1 + 1e-6) + 1e-30 or s == 262144.0             # never bolder than the exact form beyond its slack
        return min(s / HW_ULPS, 262144.0)
    return float(H.hc_pf_band_sigma_top(fp(e), thr, B, fp(box), int(b_safe), H.hc_pf_band_top(int(rule == "pack"))))


def fp(a):
    return a.ctypes.data_as(f32p)


def scales(H, thr):
    a = C.c_int(); s = [C.c_float() for _ in range(3)]
    ok = H.hc_pf_scales(thr, C.byref(a), *[C.byref(x) for x in s])
    return ok, a.value, [x.value for x in s]


def point_slots(H, X0, X1, n_real=None):
    n = X0.shape[1]
    Bn = np.zeros((n, 32), np.float32); Bt = np.zeros((n, 16), np.float32)
    for j in range(n):
        real = 1 if (n_real is None or j < n_real) else 0
        H.hc_pf_point_slots(float(X0[0, j]), float(X0[1, j]), float(X1[0, j]), float(X1[1, j]), real, fp(Bn[j]), fp(Bt[j]))
    return Bn.astype(np.float64), Bt.astype(np.float64)


def hyp_slots(H, E, thr, B, survive_all=False):
    e = np.ascontiguousarray(E, np.float32).reshape(9)
    ns = np.zeros(32, np.float32); ts = np.zeros(16, np.float32)
    c2 = H.hc_pf_hyp_slots(fp(e), thr, B, int(survive_all), fp(ns), fp(ts))
    return ns.astype(np.float64), ts.astype(np.float64), float(c2)


def rejected(H, ns, ts, Bn, Bt):
    """The rule (sign of fma(-nt, nt, G)) under the worst accumulation error: returns a bool array, True where ANY
    admissible perturbation rejects."""
    nt = Bn @ ns; G = Bt @ ts
    en = ACC * (np.abs(Bn) @ np.abs(ns)); eg = ACC * (np.abs(Bt) @ np.abs(ts))
    out = np.zeros(nt.shape, bool)
    for sn in (1.0, -1.0, 0.0):
        for sg in (1.0, -1.0, 0.0):
            n32 = (np.sign(nt) * (np.abs(nt) + sn * en)).astype(np.float32).astype(np.float64)
            g32 = (G + sg * eg).astype(np.float32).astype(np.float64)
            out |= (g32 - n32 * n32) < 0.0                       # exact in float64 (24 + 48 bits): the sign a single-rounding fma returns
    return out


def rejected_band(H, ns, Bn):
    """The band rule (bit 30 of the accumulator: |nt| >= 2) under the worst accumulation error."""
    nt = Bn @ ns
    en = ACC * (np.abs(Bn) @ np.abs(ns))
    worst = (np.abs(nt) + en).astype(np.float32)
    bits = worst.view(np.uint32)
    rej = ((bits >> 30) & 1).astype(bool)
    assert np.array_equal(rej, worst >= 2.0)                        # the bit IS the comparison (no NaN / inf among the operands)
    for k in np.flatnonzero(rej)[:4]:
        assert H.hc_pf_band_reject(float(worst[k])) == 1
    return rej


def rejected_pack(H, ns, Bn):
    """The packed scan (the six-bit conversion's top exponent bit: |nt| >= 1.875) under the worst accumulation error."""
    nt = Bn @ ns
    en = ACC * (np.abs(Bn) @ np.abs(ns))
    worst = (np.abs(nt) + en).astype(np.float32)
    rej = worst >= np.float32(1.875)
    for k in list(np.flatnonzero(rej)[:4]) + list(np.flatnonzero(~rej)[:4]):
        assert H.hc_pf_band_pack_reject(float(worst[k])) == int(rej[k]) == H.hc_pf_band_pack_reject(-float(worst[k]))
    return rej


def band_box(H, X0, X1, n, B):
    """The boxes as fill_xu_kernel leaves them: ordered-bit maxima of (x, -x, y, -y | u, -u, v, -v) over the points that carry
    features, decoded by pf_box_from_words."""
    with np.errstate(invalid="ignore"):
        c = np.stack([X0[0, :n], X0[1, :n], X1[0, :n], X1[1, :n]])
        feat = np.isfinite(c).all(axis=0) & (np.abs(c).max(axis=0) <= 48.0)
    words = (C.c_uint64 * 8)()
    vals = [X1[0, :n], -X1[0, :n], X1[1, :n], -X1[1, :n], X0[0, :n], -X0[0, :n], X0[1, :n], -X0[1, :n]]
    for k, v in enumerate(vals):
        m = np.float32(v[feat].max()) if feat.any() else np.float32(-np.inf)
        words[k] = (7 << 32) | H.hc_pf_order_bits(float(m))          # an epoch in the upper half, ignored by the decoder
    box = np.zeros(8, np.float32)
    H.hc_pf_box_from_words(words, float(B), fp(box))
    if feat.any():
        assert box[0] <= X1[0, :n][feat].min() and box[1] >= X1[0, :n][feat].max() and box[4] <= X0[0, :n][feat].min() and box[7] >= X0[1, :n][feat].max()
    return box


class ZeroDivisorGuard:
    """prefilter_math.hpp (3) as the kernel stages it: occupied cells of the tile's points, per hypothesis the 2 x 2 cell
    test, and the scan of the whole tile for those it cannot clear.  decide() also checks the guard's own claim -- a
    hypothesis it clears must not have a zero-divisor point (brute force over the tile).  side = 1: the SECOND divisor
    (transposed system, first-view positions, flipped keys) -- the band rule only asks whether it can be ruled out."""

    def __init__(self, H, X1, n, B, side=0):
        self.H, self.B = H, float(B)
        self.side = side
        ok = np.isfinite(X1[:2, :n]).all(axis=0)
        self.x = np.ascontiguousarray(X1[0, :n], np.float32); self.y = np.ascontiguousarray(X1[1, :n], np.float32)
        self.n = n
        self.keys = set()
        cell = (C.c_int * 2)(); key = C.c_uint32()
        for j in np.nonzero(ok)[0]:
            H.hc_pf_point_cell(float(self.x[j]), float(self.y[j]), self.B, cell, C.byref(key))
            self.keys.add(H.hc_pf_cell_key_side(cell[0], cell[1], side))
            assert side == 1 or self.keys.issuperset({key.value})
        self.states = [0, 0, 0]
        self.scans = 0

    def decide(self, E):
        e = np.ascontiguousarray(E, np.float32).reshape(9)
        if self.side:
            et = np.zeros(9, np.float32)
            self.H.hc_pf_transposed(fp(e), fp(et))
            e = et
        cells = (C.c_int * 4)(); g = C.c_float()
        st = self.H.hc_pf_zero_divisor_cells(fp(e), self.B, cells, C.byref(g))
        self.states[st] += 1
        with np.errstate(invalid="ignore"):
            truth = self.H.hc_pf_zero_divisor_any(fp(e), fp(self.x), fp(self.y), self.n) > 0
        if st == 1:
            assert 0 <= cells[1] - cells[0] <= 1 and 0 <= cells[3] - cells[2] <= 1
            hit = any(self.H.hc_pf_cell_key_side(cx, cy, self.side) in self.keys for cx in range(cells[0], cells[1] + 1) for cy in range(cells[2], cells[3] + 1))
            st = 2 if hit else 0
        if st == 0:
            assert not truth, "the guard cleared a hypothesis that has a zero-divisor point in the tile"
            return False
        self.scans += 1
        return truth if self.side == 0 else True                     # second divisor: "cannot rule it out" is all the rule asks


def check_scene(H, X0, X1, Es, thr, n_real=None, max_survivors=None, scans_below=None, rule="G"):
    ok, a, (sigE, sigF, sig2a) = scales(H, thr)
    assert ok
    n = X0.shape[1] if n_real is None else n_real
    with np.errstate(invalid="ignore"):
        c = np.abs(np.concatenate([X0[:2, :n].ravel(), X1[:2, :n].ravel()]))
        B = float(np.max(c[c <= 48.0], initial=0.0))
    Bn, Bt = point_slots(H, X0, X1, n_real)
    surv = 0
    guard = ZeroDivisorGuard(H, X1, n, B)
    if rule in BAND_RULES:
        guard_b = ZeroDivisorGuard(H, X0, n, B, side=1)
        box = band_box(H, X0, X1, n, B)
    for E in Es:
        if rule in BAND_RULES:
            e = np.ascontiguousarray(E, np.float32).reshape(9)
            with np.errstate(invalid="ignore", over="ignore"):
                b_safe = not guard_b.decide(E)
                sigma = 0.0 if guard.decide(E) else band_sigma(H, e, thr, B, box, b_safe, rule)
            assert 0.0 <= sigma <= 262144.0
            if rule == "band" and sigma > 0:
                assert np.float32(sigma) == np.float32(H.hc_pf_band_sigma(fp(e), thr, B, fp(box), int(b_safe)))      # the default top is round 5's
            ns32 = np.zeros(32, np.float32)
            H.hc_pf_band_hyp_slots(fp(e), sigma, fp(ns32))
            assert ns32[27] == 1.0 and (sigma > 0 or not ns32[:27].any())
            rej = (rejected_band if rule == "band" else rejected_pack)(H, ns32.astype(np.float64), Bn)
        else:
            ns, ts, _ = hyp_slots(H, E, thr, B, guard.decide(E))
            rej = rejected(H, ns, ts, Bn, Bt)
        _, mask = O.count_inliers(np.ascontiguousarray(E, np.float32).reshape(3, 3), X0[:, :n], X1[:, :n], thr)
        inl = np.zeros(rej.shape, bool); inl[:n] = mask.astype(bool)
        assert not (rej & inl).any(), f"rejected {int((rej & inl).sum())} oracle inliers"
        if n_real is not None:
            assert rej[n:].all(), "padding must always be rejected"
        surv += int((~rej[:n]).sum())
    rate = surv / (len(Es) * n)
    if max_survivors is not None:
        assert rate < max_survivors, f"survivor rate {rate:.4f}"
    if scans_below is not None:
        assert guard.scans <= scans_below * len(Es), f"{guard.scans} of {len(Es)} hypotheses needed a tile scan"
    return rate


@pytest.mark.parametrize("rule", RULES)
@pytest.mark.parametrize("thr", [1e-8, 1e-6, 1e-4, 1e-3])
@pytest.mark.parametrize("focal", [600.0, 2360.0])
def test_no_oracle_inlier_is_rejected(H, thr, focal, rule):
    n = 1024
    sc = synth.two_view_scene(n, seed=7, focal=focal)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    Es = [O.hypothesis_E(X0, X1, O.sample8(99, h, n), 0) for h in range(96)]
    rate = check_scene(H, X0, X1, Es, np.float32(thr), scans_below=0.05, rule=rule)
    if thr == 1e-6 and focal == 2360.0:
        assert rate < (0.04 if rule in BAND_RULES else 0.03)        # and it still filters: 1-2 % survive at the reference threshold


@pytest.mark.parametrize("rule", RULES)
def test_padding_and_ragged_tile(H, rule):
    n = 300
    sc = synth.two_view_scene(384, seed=3)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    X0[:, n:] = np.nan; X1[:, n:] = np.nan                          # what the tail of a padded row looks like
    Es = [O.hypothesis_E(X0[:, :n], X1[:, :n], O.sample8(5, h, n), 0) for h in range(40)]
    check_scene(H, X0, X1, Es, np.float32(1e-6), n_real=n, rule=rule)


@pytest.mark.parametrize("rule", RULES)
def test_zero_divisor_pairs_survive(H, rule):
    """da_c == 0 zeroes the first term of the residual (the reference's element_wise_div guard): forward motion with a
    correspondence whose x2 sits exactly on the epipole has r = n^2 / db, possibly an inlier, although n^2 / da is 'infinite'."""
    E = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 0]], np.float32)            # t = (0, 0, 1), R = I; singular values (1, 1, 0)
    rng = np.random.default_rng(0)
    n = 64
    X0 = np.ones((3, n), np.float32); X1 = np.ones((3, n), np.float32)
    X0[:2] = rng.uniform(-0.3, 0.3, (2, n)).astype(np.float32)
    X1[:2] = (X0[:2] * np.float32(1.05)).astype(np.float32)                  # forward motion: points move radially -> inliers
    X1[:2, 0] = 0.0                                                          # x2 on the epipole: a0 = a1 = 0 exactly
    X1[:2, 1] = 0.0; X0[:2, 1] = 0.0                                         # both epipoles: da = db = 0, r = 0
    X1[:2, 2] = np.float32(1e-30)                                            # da underflows
    cnt, mask = O.count_inliers(E, X0, X1, np.float32(1e-6))
    assert mask[0] == 1 and mask[1] == 1
    check_scene(H, X0, X1, [E, -E, (E * np.float32(0.5)).astype(np.float32)], np.float32(1e-6), rule=rule)
    # the same with n != 0: rows 0 and 1 parallel (rank 2), x2 orthogonal to them but not to row 2 -> a0 = a1 = 0 exactly,
    # n = a2 = x2y, r = n^2 / db: an inlier for |x2y| < 1e-3 that the one-sided bound n^2 / da = inf would throw away
    E2 = np.array([[1, 0, -0.125], [2, 0, -0.25], [0, 1, 0]], np.float32)
    Y0 = np.ones((3, 8), np.float32); Y1 = np.ones((3, 8), np.float32)
    Y0[:2] = rng.uniform(-0.3, 0.3, (2, 8)).astype(np.float32)
    Y1[0] = 0.125
    Y1[1] = np.array([5e-4, -9e-4, 2e-3, 0.0, 1e-5, 0.3, -0.3, 9.9e-4], np.float32)
    cnt, mask = O.count_inliers(E2, Y0, Y1, np.float32(1e-6))
    assert mask[0] == 1 and mask[1] == 1 and mask[2] == 0 and mask[5] == 0
    check_scene(H, Y0, Y1, [E2], np.float32(1e-6), rule=rule)
    # the test is live: without the zero-divisor guard the rule WOULD reject those inliers
    Bn, Bt = point_slots(H, Y0, Y1)
    if rule in BAND_RULES:
        sigma = band_sigma(H, E2.reshape(9), np.float32(1e-6), 0.3, band_box(H, Y0, Y1, 8, 0.3), 1, rule)
        ns32 = np.zeros(32, np.float32)
        H.hc_pf_band_hyp_slots(fp(E2.reshape(9)), sigma, fp(ns32))
        # (the first divisor's maximum over the box is 0 here, so the constant is the error floor alone: |n| = 5e-4 and 9e-4 are far outside)
        assert sigma > 0 and (rejected_band if rule == "band" else rejected_pack)(H, ns32.astype(np.float64), Bn)[:2].all()
    else:
        ns, ts, _ = hyp_slots(H, E2, np.float32(1e-6), 0.3, survive_all=False)
        assert rejected(H, ns, ts, Bn, Bt)[:2].all()
    g = ZeroDivisorGuard(H, Y1, 8, 0.3)
    assert g.decide(E2) and g.scans == 1


@pytest.mark.parametrize("rule", RULES)
@pytest.mark.parametrize("scale,thr", [(0.3, 1e-6), (5.0, 1e-3)])
def test_crafted_candidates_of_the_gpu_test(H, scale, thr, rule):
    """The candidate set of tests/test_gpu_prefilter.py::test_prefilter_supplied_candidates_and_zero_divisors through the
    host model of the rule (one tile of it)."""
    from helpers import crafted_candidates, lattice_points
    rng = np.random.default_rng(3)
    X0, X1 = lattice_points(rng, 1024, scale)
    Es = crafted_candidates(rng, X1, 1024, 512)
    kept = sum(O.count_inliers(Es[h], X0, X1, np.float32(thr))[0] for h in range(10, 512, 16))
    assert kept > 0
    with np.errstate(invalid="ignore", over="ignore"):
        check_scene(H, X0, X1, list(Es), np.float32(thr), rule=rule)


@pytest.mark.parametrize("rule", RULES)
def test_degenerate_hypotheses_and_points(H, rule):
    sc = synth.two_view_scene(256, seed=11)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    X0 = X0.copy(); X1 = X1.copy()
    X1[0, 5] = np.nan; X0[1, 6] = np.inf; X1[0, 7] = 100.0; X0[0, 8] = -60.0          # non-finite / beyond the fp16 feature range
    good = O.hypothesis_E(X0, X1, O.sample8(1, 3, 256), 0)
    Es = [np.zeros((3, 3), np.float32),                                                # E = 0: every finite pair has r = 0 -> inlier
          np.full((3, 3), np.nan, np.float32), good * np.float32(3.0),                 # non-finite, and entries above the tame bound
          good, np.diag([1, 1, 0]).astype(np.float32), np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)]
    with np.errstate(invalid="ignore"):
        check_scene(H, X0, X1, Es, np.float32(1e-6), rule=rule)


@pytest.mark.parametrize("rule", RULES)
def test_random_matrices_and_scales(H, rule):
    rng = np.random.default_rng(5)
    n = 512
    for scale in (0.05, 0.5, 3.0, 20.0):
        X0 = np.ones((3, n), np.float32); X1 = np.ones((3, n), np.float32)
        X0[:2] = (scale * rng.uniform(-1, 1, (2, n))).astype(np.float32)
        X1[:2] = (X0[:2] + 0.02 * scale * rng.normal(size=(2, n))).astype(np.float32)
        Es = []
        for _ in range(24):
            M = rng.normal(size=(3, 3)) * rng.choice([1e-3, 0.1, 1.0], size=(3, 3))
            M /= max(1e-9, np.abs(M).max())
            Es.append((M * rng.uniform(0.2, 1.9)).astype(np.float32))
        for thr in (1e-7, 1e-5, 1e-3):
            t = np.float32(min(max(thr * scale * scale, 1e-9), 1e-2))            # inside the range the fp16 scaling covers
            check_scene(H, X0, X1, Es, t, rule=rule)


def test_threshold_range(H):
    for thr, ok in ((1e-10, 0), (1e-9, 1), (1e-6, 1), (1e-2, 1), (0.02, 0), (float("nan"), 0)):
        assert scales(H, thr)[0] == ok
    assert scales(H, 1e-6)[1] == 7 and scales(H, 1e-6)[2] == [8.0, 16.0, 16384.0]


def exact_zero_family(rng, pts, count):
    """Matrices whose first two rows vanish EXACTLY (in float arithmetic) at one of the given points: coefficients with few
    mantissa bits and points on a 2^-12 lattice make every product and sum exact.  Well and ill conditioned 2 x 2 parts."""
    out = []
    coef = np.array([-1.5, -1.0, -0.75, -0.5, -0.25, 0.25, 0.5, 0.75, 1.0, 1.5])
    while len(out) < count:
        px, py = pts[rng.integers(len(pts))]
        a, b = rng.choice(coef, 2)
        kind = rng.integers(4)
        if kind == 0:
            c, d = rng.choice(coef, 2)
        elif kind == 1:
            c, d = a + 2.0 ** -rng.integers(4, 12), b                      # nearly parallel rows
        elif kind == 2:
            c, d = 2.0 * a, 2.0 * b                                        # parallel rows: singular, a whole line of zeros
            if abs(c) > 2 or abs(d) > 2:
                c, d = 0.5 * a, 0.5 * b
        else:
            c, d = 0.0, 0.0                                                # second row vanishes identically with b1 = 0
        e2 = -(a * px + b * py); e5 = -(c * px + d * py)
        E = np.array([[a, b, e2], [c, d, e5], rng.uniform(-1, 1, 3)], np.float64)
        if np.abs(E).max() > 2.0:
            continue
        E32 = E.astype(np.float32)
        assert np.array_equal(E32.astype(np.float64)[:2], E[:2])
        out.append(E32)
    return out


def test_zero_divisor_guard_claims(H):
    """The per-(hypothesis, tile) guard on its own: whenever it clears a hypothesis (no cell hit), brute force must find no
    point with da_c == 0 (asserted inside decide); crafted exact zeros must be found; random matrices hardly ever scan."""
    rng = np.random.default_rng(12)
    n = 1024
    for scale in (0.3, 1.0, 7.0, 40.0):
        lattice = 2.0 ** -12 * 2.0 ** np.ceil(np.log2(scale))
        X1 = np.ones((3, n), np.float32)
        X1[:2] = (np.round(rng.uniform(-scale, scale, (2, n)) / lattice) * lattice).astype(np.float32)
        X1[0, 5] = np.nan; X1[1, 6] = np.inf
        B = float(np.abs(X1[:2][np.isfinite(X1[:2])]).max())
        guard = ZeroDivisorGuard(H, X1, n, B)
        pts = [(float(X1[0, j]), float(X1[1, j])) for j in range(8, 200)]
        found = 0
        fam = exact_zero_family(rng, pts, 300)
        for E in fam:
            found += bool(guard.decide(E))
        assert found == len(fam), "every crafted matrix has an exact zero divisor in the tile"
        # the same matrices moved off the lattice: the zero is (almost always) gone, the claim still has to hold
        for E in fam[:150]:
            E2 = E.copy(); E2[0, 2] = np.nextafter(E2[0, 2], np.float32(9), dtype=np.float32)
            guard.decide(E2)
        scans0 = guard.scans
        for _ in range(1500):
            M = rng.normal(size=(3, 3)) * rng.choice([1e-3, 0.05, 1.0], size=(3, 3))
            M *= rng.uniform(0.1, 1.9) / max(1e-9, np.abs(M).max())
            assert not guard.decide(M.astype(np.float32)) or True
        assert guard.scans - scans0 < 0.6 * 1500                          # badly scaled on purpose; real hypotheses: test_no_oracle_inlier_is_rejected (< 5 %)
    # degenerate inputs: all-zero, zero 2 x 2 part with and without a constant, non-finite entries
    X1 = np.ones((3, 64), np.float32); X1[:2] = rng.uniform(-0.3, 0.3, (2, 64)).astype(np.float32)
    guard = ZeroDivisorGuard(H, X1, 64, 0.3)
    assert guard.decide(np.zeros((3, 3), np.float32))
    assert not guard.decide(np.array([[0, 0, 0.5], [0, 0, 0], [1, 0, 0]], np.float32))
    assert guard.decide(np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0]], np.float32))
    assert guard.decide(np.array([[0, 0, 1e-30], [0, 0, 1e-30], [1, 0, 0]], np.float32))       # the squares underflow
    with np.errstate(invalid="ignore"):
        assert not guard.decide(np.full((3, 3), np.nan, np.float32))


def test_committed_tie_cases_are_kept_and_their_splits_are_consistent(H):
    """tests/golden/prefilter_tie_cases.json: oracle inliers whose x2x^2 is an exact fp16 half-way case in fp32 (the device
    build once split such a feature inconsistently; GPU twin: test_prefilter_keeps_inliers_whose_feature_is_an_fp16_tie)."""
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prefilter_tie_cases.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        X0 = np.float32([[c["x1"][0], -1.75], [c["x1"][1], 1.06], [1.0, 1.0]])
        X1 = np.float32([[c["x2"][0], 1.7], [c["x2"][1], -1.34], [1.0, 1.0]])
        E = np.float32(c["E"]).reshape(3, 3)
        cnt, mask = O.count_inliers(E, X0, X1, np.float32(c["thr"]))
        assert mask[0]
        check_scene(H, X0, X1, [E], float(np.float32(c["thr"])))
        check_scene(H, X0, X1, [E], float(np.float32(c["thr"])), rule="band")
        check_scene(H, X0, X1, [E], float(np.float32(c["thr"])), rule="pack")
        p = np.float32(X1[0, 0] * X1[0, 0])
        assert abs(float(p) - float(np.float16(p))) * 2 == float(np.spacing(np.float16(p)))         # the tie
        Bn, Bt = point_slots(H, X0[:, :1], X1[:, :1])
        assert Bt[0][0] == Bt[0][2] and abs(Bt[0][0] + Bt[0][1] - float(p)) <= 2.0 ** -21 * float(p)


# ---- band rule only ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rule", BAND_RULES)
def test_band_second_divisor_zero_selects_the_weaker_constant(H, rule):
    """db_c == 0 zeroes the SECOND term: the pair is an inlier iff n^2 < thr da, which can exceed the harmonic constant
    thr Da Db / (Da + Db) the rule uses when no point can have db_c == 0.  The guard on the transposed system must notice."""
    thr = np.float32(1e-6)
    # b = (e0 u + e3 v + e6, e1 u + e4 v + e7) vanishes exactly at x1 = (0.25, -0.125); a = (x, y) + (0.5, 0.25): |a| ~ 0.56
    E = np.array([[1, 0, 0.5], [0, 1, 0.25], [-0.25, 0.125, 0.0]], np.float32)
    n = 16
    rng = np.random.default_rng(4)
    X0 = np.ones((3, n), np.float32); X1 = np.ones((3, n), np.float32)
    X0[:2] = rng.uniform(-0.3, 0.3, (2, n)).astype(np.float32)
    X1[:2] = rng.uniform(-0.3, 0.3, (2, n)).astype(np.float32)
    X0[0, 0] = 0.25; X0[1, 0] = -0.125                                   # db_c == 0 for point 0
    # with b(x1) = 0 the numerator is n = b2 = e2 u + e5 v + e8 = 0.09375 + e8 whatever x2 is; x2 in the corner of the box where da is
    # largest (0.94, against a harmonic constant of ~0.32): e8 makes n^2 ~ 0.8 thr da
    X1[0, 0] = 0.3; X1[1, 0] = 0.3
    E[2, 2] = np.float32(-0.09375 + 8.67e-4)
    r = O.residual(E, (0.25, -0.125, 1.0), (float(X1[0, 0]), float(X1[1, 0]), 1.0))
    assert 0.6 * thr < r < 0.95 * thr                                     # r = n^2 / da alone: the second term is zeroed
    cnt, mask = O.count_inliers(E, X0, X1, thr)
    assert mask[0] == 1
    check_scene(H, X0, X1, [E], thr, rule=rule)
    # live: with the harmonic constant (b_safe forced) the same pair is rejected
    Bn, _ = point_slots(H, X0, X1)
    box = band_box(H, X0, X1, n, 0.3)
    for b_safe, expect in ((1, True), (0, False)):
        sigma = band_sigma(H, E.reshape(9), thr, 0.3, box, b_safe, rule)
        ns32 = np.zeros(32, np.float32)
        H.hc_pf_band_hyp_slots(fp(E.reshape(9)), sigma, fp(ns32))
        assert bool((rejected_band if rule == "band" else rejected_pack)(H, ns32.astype(np.float64), Bn)[0]) == expect


@pytest.mark.parametrize("rule", BAND_RULES)
def test_band_small_and_large_matrices_sigma_clamp(H, rule):
    """Tiny first rows make the constant tiny and sigma hits its fp16-range clamp (2^18); the rule must stay conservative."""
    rng = np.random.default_rng(8)
    n = 512
    sc = synth.two_view_scene(n, seed=21)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    base = [O.hypothesis_E(X0, X1, O.sample8(3, h, n), 0) for h in range(12)]
    Es = []
    for E in base:
        for f in (1e-7, 1e-5, 1e-3, 1.0, 1.9):
            Es.append((E * np.float32(f)).astype(np.float32))
        M = E.copy(); M[:2] *= np.float32(1e-6); Es.append(M)              # divisors ~1e-12, n of order one
        M = E.copy(); M[2] *= np.float32(1e-6); Es.append(M)
    clamped = 0
    box = band_box(H, X0, X1, n, float(np.abs(np.concatenate([X0[:2].ravel(), X1[:2].ravel()])).max()))
    for E in Es:
        clamped += band_sigma(H, np.ascontiguousarray(E, np.float32).reshape(9), np.float32(1e-6), 0.7, box, 1, rule) == 262144.0
    assert clamped >= 12
    for thr in (1e-9, 1e-6, 1e-2):
        check_scene(H, X0, X1, Es, np.float32(thr), rule=rule)


def test_band_ordered_bits_and_boxes(H):
    vals = np.float32([-np.inf, -48.0, -1.0, -1e-30, -0.0, 0.0, 1e-30, 0.5, 48.0, np.inf])
    keys = [H.hc_pf_order_bits(float(v)) for v in vals]
    assert keys == sorted(keys) and len(set(keys)) == len(keys) - 0 or keys[4] < keys[5]      # -0 < +0 in the key order, everything else strictly increasing
    # a side without points (all maxima -inf) falls back to [-B, B]; so does a range that leaves it
    words = (C.c_uint64 * 8)(*[H.hc_pf_order_bits(float("-inf"))] * 8)
    box = np.zeros(8, np.float32)
    H.hc_pf_box_from_words(words, 0.5, fp(box))
    assert np.array_equal(box, np.float32([-0.5, 0.5, -0.5, 0.5, -0.5, 0.5, -0.5, 0.5]))
    m = [0.25, 0.125, 0.3, -0.1, 0.6, 0.2, 0.1, 0.1]                     # maxima of (x, -x, y, -y, u, -u, v, -v): u's 0.6 exceeds B
    words = (C.c_uint64 * 8)(*[(3 << 32) | H.hc_pf_order_bits(v) for v in m])
    H.hc_pf_box_from_words(words, 0.5, fp(box))
    assert np.allclose(box, [-0.125, 0.25, 0.1, 0.3, -0.5, 0.5, -0.5, 0.5])
    # the words as they lie behind the bound: boxes of the bound's own fillXU epoch are taken, ONE word of another epoch (no cell pass for these
    # points yet, or a failed one) sends both views back to [-B, B] -- a stale, smaller box would make sigma too large and reject inliers
    m = [0.25, 0.125, 0.3, -0.1, 0.4, 0.2, 0.1, 0.1]
    B = np.float32(0.5)
    bound = (C.c_uint64 * 10)(*([(7 << 32) | int(B.view(np.uint32)), 0] + [(7 << 32) | H.hc_pf_order_bits(v) for v in m]))
    H.hc_pf_box_from_bound(bound, float(B), fp(box))
    assert np.allclose(box, [-0.125, 0.25, 0.1, 0.3, -0.2, 0.4, -0.1, 0.1])
    for k in range(8):
        stale = (C.c_uint64 * 10)(*bound)
        stale[2 + k] = (6 << 32) | H.hc_pf_order_bits(m[k])
        H.hc_pf_box_from_bound(stale, float(B), fp(box))
        assert np.array_equal(box, np.float32([-0.5, 0.5] * 4)), k


def test_band_survivor_rate_on_the_bench_scene(H):
    """What the rule costs: ~1.2 % of the pairs of the bench scene survive (G rule: ~0.7 %, inliers: 0.35 %)."""
    n = 1024
    sc = synth.two_view_scene(4096)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    X0 = np.ascontiguousarray(X0[:, :n]); X1 = np.ascontiguousarray(X1[:, :n])
    Es = [O.hypothesis_E(X0, X1, O.sample8(0x5EED5F3D, h, n), 0) for h in range(200)]
    rate_b = check_scene(H, X0, X1, Es, np.float32(1e-6), rule="band")
    rate_p = check_scene(H, X0, X1, Es, np.float32(1e-6), rule="pack")
    rate_g = check_scene(H, X0, X1, Es, np.float32(1e-6), rule="G")
    assert rate_g < rate_b < 0.02 and abs(rate_p - rate_b) < 0.002 * rate_b + 1e-4          # the scans cut at 2 / 1.998 and 1.875 / 1.873 of the same band


def test_pack_survivor_table_against_a_model_of_the_conversion(H):
    """pf_pack_code: v_cvt_scalef32_2xpk16_bf6_f32 writes accumulator j of step s as six-bit field f = 2 j + s (bits 6 f .. 6 f + 5 of
    six registers; layout and rounding measured on the MI355X: profiles/r06_cvt_pack_probe.txt), the kernel merges the top exponent
    bits (6 f + 4) of registers 0..2 into the even bits of a word and those of registers 3..5 into the odd ones.  Rebuilt here
    with numpy for every single-survivor pattern; the table must name the slot.  (GPU twin: the band probe's pack_slots_ok.)"""
    M0, M1, M2 = 0x10410410, 0x04104104, 0x41041041
    assert M0 | M1 | M2 == 0x55555555 and M0 & M1 == 0 and M0 & M2 == 0 and M1 & M2 == 0
    seen = set()
    for j in range(16):
        for s in range(2):
            bits = 0                                     # 192-bit string: every field rejected (top exponent bit set) but (j, s)
            for f in range(32):
                if f != 2 * j + s:
                    bits |= 1 << (6 * f + 4)
                bits |= (f * 7 % 16) << (6 * f)           # rubbish in the mantissa / low exponent bits, and the sign
                bits |= (f % 2) << (6 * f + 5)
            d = [(bits >> (32 * r)) & 0xFFFFFFFF for r in range(6)]
            wa = (d[0] & M0) | (d[1] & M1) | (d[2] & M2)
            wb = (d[3] & M0) | (d[4] & M1) | (d[5] & M2)
            rej = (wa & 0x55555555) | ((wb << 1) & 0xAAAAAAAA)
            surv = ~rej & 0xFFFFFFFF
            assert bin(surv).count("1") == 1
            b = 32 - surv.bit_length()                   # count of leading zeros
            assert H.hc_pf_pack_code(b) == ((j & 3) + 8 * (j >> 2)) | (s << 5), (j, s, b)
            assert H.hc_pf_pack_field(b) == 2 * j + s, (j, s, b)          # the form the kernel runs (a shift out of a 64-bit table)
            seen.add(b)
    assert seen == set(range(32))
    assert H.hc_pf_band_pack_reject(1.875) == 1 and H.hc_pf_band_pack_reject(float(np.nextafter(np.float32(1.875), np.float32(0)))) == 0
    assert 1.87 < H.hc_pf_band_top(1) < 1.875 and 1.99 < H.hc_pf_band_top(0) < 2.0


def test_per_tile_boxes_in_morton_order_never_reject_an_inlier(H):
    """The per-tile rule (kPfRuleBandTile, the product): the scoring tiles are runs of the Morton-bucket order of the first view's
    positions and the band's constant is a maximum over the TILE's boxes.  Every tile is a scene of its own for the rule (check_scene
    takes the boxes and the bound from the points it is given: the tile's own, smaller bound is the stricter test), so no oracle
    inlier may be rejected in any tile -- and the tiles' survivor rate is below the whole-view rule's."""
    n, tile = 4096, 1024
    sc = synth.two_view_scene(n)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    X0 = np.ascontiguousarray(X0[:, :n]); X1 = np.ascontiguousarray(X1[:, :n])
    ulo, uhi, vlo, vhi = [float(f) for f in (X0[0].min(), X0[0].max(), X0[1].min(), X0[1].max())]
    keys = np.array([H.hc_pf_morton_key(float(X0[0, j]), float(X0[1, j]), ulo, uhi, vlo, vhi) for j in range(n)], np.uint64)
    order = np.argsort(keys >> np.uint64(20), kind="stable")          # the device's bucket ordering: the top ten bits of the key, the original order inside a bucket
    Es = [O.hypothesis_E(X0, X1, O.sample8(0x5EED5F3D, h, n), 0) for h in range(60)]
    thr = np.float32(1e-6)
    whole = check_scene(H, X0[:, :tile], X1[:, :tile], Es, thr, rule="pack")      # one unsorted tile's worth of points with ITS boxes (~ the whole view's)
    rates, areas = [], []
    assert keys.max() < 2 ** 30
    for t0 in range(0, n, tile):
        ids = order[t0:t0 + tile]
        a = np.ascontiguousarray(X0[:, ids]); b = np.ascontiguousarray(X1[:, ids])
        areas.append(float(np.ptp(a[0]) * np.ptp(a[1])))
        rates.append(check_scene(H, a, b, Es, thr, rule="tile"))
    assert np.mean(areas) < 0.7 * (uhi - ulo) * (vhi - vlo)                          # the order does what it is for: smaller first-view boxes
    assert np.mean(rates) < whole
