"""GPU tests of the LAB-BENCH flavour of the library (libsfm_amd_ab.so = the product's sources built with -DSFM_AB=1,
include/sfm_amd_ab.h): the A/B switches behind sfm_ransac_params.reserved[], the recorded slower kernel variants (f32
matrix-core scoring, the round-2 pre-filter kernel, the generic lane-solve kernel) and the probe hook -- each against the
oracle, like the product's kernels.  And the other side of the split: the PRODUCT library refuses all of it."""
import os

import numpy as np
import pytest

import cuda_sfm_amd as PROD
import cuda_sfm_amd_ab as S
from cuda_sfm_amd_ab import synth
import oracle as O
from helpers import same_bits, to_dev, make_pair
import test_gpu_prefilter as P
import test_gpu_ransac as R

pytestmark = pytest.mark.gpu


def test_flavours():
    assert S.AB and not PROD.AB and S.LIB_PATH != PROD.LIB_PATH
    assert S.lib().sfm_abi_version() == PROD.lib().sfm_abi_version() == 3
    for name in S.AB_EXPORTS:
        assert hasattr(S.lib(), name) and not hasattr(PROD.lib(), name), name


def test_product_library_refuses_the_switches(gpu):
    """libsfm_amd.so: non-zero reserved[] and kernel id 3 are SFM_E_INVALID -- before anything is launched."""
    n = 256
    scene = synth.two_view_scene(n, seed=1)
    pair, _ = make_pair(PROD, gpu, scene)
    for i in range(4):
        p = PROD.default_params(n, num_hypotheses=64)
        p.reserved[i] = 1
        with pytest.raises(PROD.SfmError) as e:
            pair.estimateE(p)
        assert e.value.code == PROD.E_INVALID and "reserved" in str(e.value)
        with pytest.raises(PROD.SfmError):
            pair.ransac_score(p)
    p = PROD.default_params(n, num_hypotheses=64, kernel=3)
    with pytest.raises(PROD.SfmError) as e:
        pair.estimateE(p)
    assert e.value.code == PROD.E_INVALID
    pair.estimateE(PROD.default_params(n, num_hypotheses=64))           # and the pair is still usable
    assert pair.get_best()[1] > 0


@pytest.mark.parametrize("sweeps", [0, 7])
@pytest.mark.parametrize("n,H", [(64, 50), (1000, 300), (4096, 2048), (4500, 600), (9000, 100), (129, 65), (511, 8193)])
def test_mfma_scoring_counts_winner_mask_E(gpu_ab, n, H, sweeps):
    """SFM_KERNEL_MFMA (csrc/ab/ransac_mfma.hip): E.X on the f32 matrix cores, every count / key / E / mask against the oracle."""
    scene = synth.two_view_scene(n, seed=5 + n)
    pair, _ = make_pair(S, gpu_ab, scene)
    p = S.default_params(n, num_hypotheses=H, seed=77, kernel=S.KERNEL_MFMA, jacobi_sweeps=sweeps)
    pair.estimateE(p)
    assert pair.last_launch()["kernel"] == S.KERNEL_MFMA
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    key, ocounts, oE = O.ransac_range(X0, X1, 0, H, p.threshold, sweeps, seed=77, want_E=True)
    assert np.array_equal(pair.get_inlier_counts(H), ocounts) and pair.get_key() == key
    ocnt, ohyp = O.unpack_key(key)
    assert same_bits(pair.get_E(), oE[ohyp].reshape(3, 3))
    assert np.array_equal(pair.get_inlier_mask(), O.count_inliers(oE[ohyp], X0, X1, p.threshold)[1])


def test_mfma_score_into_leaves_the_key_in_caller_memory(gpu_ab):
    torch, dev, ctx = gpu_ab
    n, H = 1200, 2000
    scene = synth.two_view_scene(n, seed=71)
    pair, _ = make_pair(S, gpu_ab, scene)
    p = S.default_params(n, num_hypotheses=H, seed=4, kernel=S.KERNEL_MFMA)
    pair.ransac_score(p)
    want = pair.get_key()
    out = torch.full((1,), 5, dtype=torch.int64, device=dev)
    pair.ransac_score(S.default_params(n, num_hypotheses=H, seed=4, kernel=S.KERNEL_MFMA), key_out=out)
    torch.cuda.synchronize()
    assert int(out.cpu().numpy().view(np.uint64)[0]) == want


@pytest.mark.parametrize("n,H", [(4097, 9000), (8192, 20000), (12345, 8192), (16384, 65536)])
def test_tile_parallel_grid_equals_tile_loop(gpu_ab, n, H):
    """n > 4096: tile-parallel scoring (default) against the in-block tile loop (reserved[1] = 1) and the oracle."""
    scene = synth.two_view_scene(n, seed=n)
    pair, _ = make_pair(S, gpu_ab, scene)
    p = S.default_params(n, num_hypotheses=H, seed=3, kernel=S.KERNEL_SPLIT)
    pair.estimateE(p)
    a = (pair.get_inlier_counts(H).copy(), pair.get_key(), pair.get_inlier_mask().copy())
    q = S.default_params(n, num_hypotheses=H, seed=3, kernel=S.KERNEL_SPLIT)
    q.reserved[1] = 1
    pair.estimateE(q)
    assert np.array_equal(pair.get_inlier_counts(H), a[0]) and pair.get_key() == a[1] and np.array_equal(pair.get_inlier_mask(), a[2])
    _, _, X0, X1 = R.oracle_xu(scene)
    rng = np.random.default_rng(n)
    R._oracle_sample_check(X0, X1, p, a[0], rng.integers(0, H, 30).tolist(), n)


@pytest.mark.parametrize("n,H", [(1000, 3000), (4096, 20000), (700, 300001)])
def test_lane_solve_variants_give_the_oracle_candidates(gpu_ab, n, H):
    """The lane-solve kernel's arrangements -- one hypothesis per lane with the sampled points gathered as 16-byte records (the
    default after fillXU), two per lane (packed), scattered dword gathers, the solver-agnostic scalar kernel; with fillXU's
    unit-z points and with sfm_set_points (no records) -- must all produce the oracle's E for every hypothesis, bit for bit."""
    torch, dev, ctx = gpu_ab
    scene = synth.two_view_scene(n, seed=300 + n)
    pair, _ = make_pair(S, gpu_ab, scene)
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    Hs = min(H, 2048)                                           # oracle candidates for the head and the tail of the range
    _, _, e_head = O.ransac_range(X0, X1, 0, Hs, 1e-6, 0, seed=9, want_E=True)
    _, _, e_tail = O.ransac_range(X0, X1, H - Hs, Hs, 1e-6, 0, seed=9, want_E=True)
    cands = {}
    for variant in (0, 2, 3, 4, 1):
        p = S.default_params(n, num_hypotheses=H, seed=9, kernel=S.KERNEL_SPLIT)
        p.reserved[0] = variant
        pair.ransac_score(p)
        cands[variant] = pair.get_E_candidates(H).reshape(H, 9).copy()
        assert same_bits(cands[variant][:Hs], e_head.reshape(Hs, 9)) and same_bits(cands[variant][H - Hs:], e_tail.reshape(Hs, 9)), variant
    for variant in (2, 3, 4, 1):
        assert same_bits(cands[variant], cands[0]), variant
    # no records: pre-normalised points through sfm_set_points (generic z), same coordinates
    d0, d1 = to_dev(torch, dev, np.ascontiguousarray(X0[:, :n])), to_dev(torch, dev, np.ascontiguousarray(X1[:, :n]))
    pair.set_points(d0, d1)
    for variant in (0, 2, 3):
        p = S.default_params(n, num_hypotheses=H, seed=9, kernel=S.KERNEL_SPLIT)
        p.reserved[0] = variant
        pair.ransac_score(p)
        assert same_bits(pair.get_E_candidates(H).reshape(H, 9), cands[0]), ("set_points", variant)

@pytest.mark.parametrize("n,H", [(900, 2500), (4096, 70001)])
def test_jacobi_lane_solve_variants_give_the_oracle_candidates(gpu_ab, n, H):
    """The normal-equations + Jacobi solver (jacobi_sweeps = 7) above the fused kernel's range: one hypothesis per lane with the
    registers capped at 256 (the product's arrangement), two per lane packed (the product up to round 4), one per lane unconstrained
    and capped at 168 (spills) -- every hypothesis' E equal to the oracle's, bit for bit."""
    torch, dev, ctx = gpu_ab
    scene = synth.two_view_scene(n, seed=500 + n)
    pair, _ = make_pair(S, gpu_ab, scene)
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    Hs = min(H, 1024)
    _, _, e_head = O.ransac_range(X0, X1, 0, Hs, 1e-6, 7, seed=11, want_E=True)
    _, _, e_tail = O.ransac_range(X0, X1, H - Hs, Hs, 1e-6, 7, seed=11, want_E=True)
    cands = {}
    for variant in (0, 2, 1, 7):
        p = S.default_params(n, num_hypotheses=H, seed=11, kernel=S.KERNEL_SPLIT)
        p.jacobi_sweeps = 7
        p.reserved[0] = variant
        pair.ransac_score(p)
        cands[variant] = pair.get_E_candidates(H).reshape(H, 9).copy()
        assert same_bits(cands[variant][:Hs], e_head.reshape(Hs, 9)) and same_bits(cands[variant][H - Hs:], e_tail.reshape(Hs, 9)), variant
    for variant in (2, 1, 7):
        assert same_bits(cands[variant], cands[0]), variant


def test_prefilter_operands_on_the_device_equal_the_host_build(gpu_ab):
    """sfm_prefilter_probe: the fp16 coefficient and feature slots the device builds for one (hypothesis, point) pair, against
    tests/hostcheck (the same header compiled for the host) bit for bit -- random pairs, the committed tie cases, crafted
    exact-tie coordinates -- and hi + lo of every feature must reproduce the fp32 feature to 2^-21."""
    import ctypes as C
    import test_hostcheck_prefilter as T
    torch, dev, ctx = gpu_ab
    h = C.CDLL(T.LIB)
    f32p = O.f32p
    h.hc_pf_hyp_slots.restype = C.c_float
    h.hc_pf_hyp_slots.argtypes = [f32p, C.c_float, C.c_float, C.c_int, f32p, f32p]
    h.hc_pf_point_slots.argtypes = [C.c_float] * 4 + [C.c_int, f32p, f32p]
    rng = np.random.default_rng(77)
    pairs = [(np.float32(c["E"]), np.float32(c["thr"]), np.float32(c["x1"] + c["x2"])) for c in P._tie_cases()]
    ties = P._fp16_tie_coordinates(rng, 24)
    for k in range(24):
        E = rng.standard_normal(9).astype(np.float32); E /= np.linalg.norm(E)
        pt = rng.uniform(-1.8, 1.8, 4).astype(np.float32)
        pt[2] = ties[k]
        if k % 2:
            pt[3] = ties[(k + 5) % 24]
        pairs.append((E, np.float32(10.0 ** rng.uniform(-8, -3)), pt))
    for E, thr, pt in pairs:
        B = float(np.abs(pt).max()) * 1.01
        dv = ctx.prefilter_probe(E, thr, B, pt)
        ns, ts, _ = T.hyp_slots(h, E, float(thr), float(np.float32(B)))
        Bn, Bt = T.point_slots(h, np.float32([[pt[0]], [pt[1]], [1.0]]), np.float32([[pt[2]], [pt[3]], [1.0]]))
        for name, a, b in (("ns", dv["ns"], ns), ("ts", dv["ts"], ts), ("bn", dv["bn"], Bn[0]), ("bt", dv["bt"], Bt[0])):
            assert np.array_equal(a.astype(np.float64), b), (name, a, b)
        x, y = np.float64(pt[2]), np.float64(pt[3])
        for j, f in enumerate((np.float32(pt[2] * pt[2]), np.float32(pt[2] * pt[3]), np.float32(pt[3] * pt[3]), pt[2], pt[3])):
            hi, lo = np.float64(dv["bt"][3 * j]), np.float64(dv["bt"][3 * j + 1])
            assert dv["bt"][3 * j + 2] == dv["bt"][3 * j]
            assert abs(hi + lo - np.float64(f)) <= abs(np.float64(f)) * 2.0 ** -21 + 2.0 ** -25, (j, hi, lo, f)
        # the matrix cores on these operands: the contraction in float64 within the accumulation budget
        nt = float(Bn[0] @ ns); G = float(Bt[0] @ ts)
        assert abs(float(dv["nt"]) - nt) <= T.ACC * float(np.abs(Bn[0]) @ np.abs(ns)) + 1e-12
        assert abs(float(dv["G"]) - G) <= T.ACC * float(np.abs(Bt[0]) @ np.abs(ts)) + 1e-12


@pytest.mark.parametrize("pack", [True, False])
def test_band_rule_operands_on_the_device_equal_the_host_build(gpu_ab, pack):
    """sfm_prefilter_band_probe: sigma, the coefficient slots read back through the record the scoring kernel reads, the feature
    slots and the matrix cores' nt for one (hypothesis, point) pair, against tests/hostcheck bit for bit (sigma included: sqrtf and
    the division are correctly rounded on both sides); the rule's bit equals |nt| >= 2 (round 5's v_alignbit scan) or what the
    six-bit conversion says, |nt| >= 1.875 (round 6's packed scan, the product)."""
    import ctypes as C
    import test_hostcheck_prefilter as T
    torch, dev, ctx = gpu_ab
    h = C.CDLL(T.LIB)
    f32p = O.f32p
    h.hc_pf_band_sigma_top.restype = C.c_float
    h.hc_pf_band_sigma_top.argtypes = [f32p, C.c_float, C.c_float, f32p, C.c_int, C.c_float]
    h.hc_pf_band_top.restype = C.c_float
    h.hc_pf_band_top.argtypes = [C.c_int]
    h.hc_pf_band_hyp_slots.argtypes = [f32p, C.c_float, f32p]
    h.hc_pf_point_slots.argtypes = [C.c_float] * 4 + [C.c_int, f32p, f32p]
    h.hc_pf_zero_divisor_cells.argtypes = [f32p, C.c_float, C.POINTER(C.c_int), f32p]
    h.hc_pf_transposed.argtypes = [f32p, f32p]
    rng = np.random.default_rng(78)
    ties = P._fp16_tie_coordinates(rng, 24)
    rejected = 0
    for k in range(64):
        E = rng.standard_normal(9).astype(np.float32); E /= np.linalg.norm(E)
        if k % 7 == 0:
            E[:6] *= np.float32(10.0 ** rng.uniform(-7, -1))                        # small divisors: sigma towards its clamp
        pt = rng.uniform(-1.8, 1.8, 4).astype(np.float32)
        if k < 24:
            pt[2] = ties[k]
        thr = np.float32(10.0 ** rng.uniform(-8.5, -2.1))
        B = float(np.float32(np.abs(pt).max() * 1.01))
        lo = np.sort(rng.uniform(-B, B, (4, 2)).astype(np.float32), axis=1)
        box = np.float32([lo[0, 0], lo[0, 1], lo[1, 0], lo[1, 1], lo[2, 0], lo[2, 1], lo[3, 0], lo[3, 1]])
        if k % 5 == 0:                                                             # a point of the hypothesis' own zero band: |n| small
            a = E.reshape(3, 3) @ np.float32([pt[2], pt[3], 1.0])
            pt[0] = np.float32(-(a[1] * pt[1] + a[2]) / a[0]) if abs(a[0]) > 1e-3 else pt[0]
            B = float(np.float32(max(B, abs(pt[0]) * 1.01)))
        b_safe = k % 3 != 0
        dv = ctx.prefilter_band_probe(E, thr, B, box, b_safe, pt, pack=pack)
        sigma = float(h.hc_pf_band_sigma_top(T.fp(E), thr, B, T.fp(box), int(b_safe), h.hc_pf_band_top(int(pack))))
        assert np.float32(sigma).view(np.uint32) == np.float32(dv["sigma"]).view(np.uint32), (k, sigma, dv["sigma"])
        ns = np.zeros(32, np.float32)
        h.hc_pf_band_hyp_slots(T.fp(E), sigma, T.fp(ns))
        Bn, _ = T.point_slots(h, np.float32([[pt[0]], [pt[1]], [1.0]]), np.float32([[pt[2]], [pt[3]], [1.0]]))
        assert np.array_equal(dv["ns"], ns), (k, dv["ns"], ns)
        assert np.array_equal(dv["bn"].astype(np.float64), Bn[0])
        nt = float(Bn[0] @ ns.astype(np.float64))
        assert abs(float(dv["nt"]) - nt) <= T.ACC * float(np.abs(Bn[0]) @ np.abs(ns.astype(np.float64))) + 1e-12
        assert dv["rejected"] == (abs(float(dv["nt"])) >= (1.875 if pack else 2.0))
        rejected += dv["rejected"]
        assert dv["pack_slots_ok"] == 32 and dv["pack_bit"] == (abs(float(pt[0])) >= 1.875)
        cells = (C.c_int * 4)(); g = C.c_float()
        et = np.zeros(9, np.float32); h.hc_pf_transposed(T.fp(E), T.fp(et))
        assert dv["zero_divisor_state"] == h.hc_pf_zero_divisor_cells(T.fp(E), B, cells, C.byref(g))
        assert dv["second_divisor_state"] == h.hc_pf_zero_divisor_cells(T.fp(et), B, cells, C.byref(g))
    assert 8 < rejected < 64
    # survive-all: all-zero coefficients but the pad marker's, nt = 0
    dv = ctx.prefilter_band_probe(E, thr, B, box, True, pt, survive_all=True, pack=pack)
    assert dv["sigma"] == 0 and not dv["ns"][:27].any() and dv["ns"][27] == 1 and dv["nt"] == 0 and not dv["rejected"]


def test_packed_scan_switches_at_1_875_in_every_slot(gpu_ab):
    """The instruction the packed scan rests on, on this GPU: v_cvt_scalef32_2xpk16_bf6_f32 rounds to nearest even and saturates, so
    the top exponent bit of a field is set exactly from |x| = 1.875 up (prefilter_math.hpp, kPfBandTopPack) -- for positive and
    negative values, subnormals, huge values and infinities -- in each of the 32 (accumulator, step) slots, and the kernel's bit
    picking + survivor table name the slot (pack_slots_ok)."""
    torch, dev, ctx = gpu_ab
    E = np.float32([0, -1, 0, 1, 0, 0, 0, 0, 0]); box = np.float32([-1, 1, -1, 1, -1, 1, -1, 1])
    sw = np.float32(1.875)
    vals = [0.0, 1e-40, 1e-30, 0.25, 1.0, 1.5, 1.75, 1.86, 1.873, 1.8749, float(np.nextafter(sw, np.float32(0))), 1.875,
            float(np.nextafter(sw, np.float32(4))), 1.876, 1.9, 1.9375, 1.998, 2.0, 2.5, 3.9, 7.0, 28.0, 29.0, 255.0, 256.0, 1e10, 3e38, float("inf")]
    rng = np.random.default_rng(6)
    vals += list(rng.uniform(1.8, 1.95, 40).astype(np.float32)) + list((10.0 ** rng.uniform(-8, 8, 30)).astype(np.float32))
    for v in vals:
        for sgn in (1.0, -1.0):
            u = np.float32(sgn * v)
            dv = ctx.prefilter_band_probe(E, np.float32(1e-6), 300.0, box, True, np.float32([u, 0.1, 0.2, 0.3]), pack=True)
            assert dv["pack_bit"] == bool(abs(u) >= sw), (u, dv["pack_bit"])
            assert dv["pack_slots_ok"] == 32, (u, dv["pack_slots_ok"])


@pytest.mark.parametrize("n,H,thr", [(4096, 65536, 1e-6), (1000, 20000, 1e-4), (700, 16385, 1e-8)])
def test_round5_alignbit_scan_still_equals_oracle(gpu_ab, n, H, thr):
    """Round 5's scan of the band rule (one v_alignbit_b32 per pair, sigma = 1.998 / W), kept behind reserved[3] = 5 for A/B runs
    against the packed scan: every count, key, E, mask."""
    scene = synth.two_view_scene(n, seed=9)
    pair, _ = make_pair(S, gpu_ab, scene)
    p = S.default_params(n, num_hypotheses=H, seed=4, kernel=S.KERNEL_PREFILTER, threshold=thr)
    p.reserved[3] = 5
    pair.estimateE(p)
    assert pair.last_launch()["kernel"] == S.KERNEL_PREFILTER
    P.check_all(pair, scene, p, H, n)


@pytest.mark.parametrize("var", [16, 17])                 # 16 = the product's accumulator epilogue, 17 = round 3's tickets
@pytest.mark.parametrize("cols,launches,H", [(64, 600, 1 << 18), (1, 150, 1 << 18)])
def test_prefilter_tickets_under_contention_forced_columns(gpu_ab, cols, launches, H, var):
    """The arg-max of the scoring kernel rests on an ordering assumption (ransac_prefilter.hip: the count atomics of a
    wavefront are acknowledged -- s_waitcnt vmcnt(0) -- before its ticket is issued, and the wavefront that draws the last
    ticket of a group then reads final counts), not on a release / acquire fence (which costs an L2 write-back per group).
    This test exercises it: 16 tiles (so 16 wavefronts on 16 CUs race for every group's tickets), grid columns forced to
    1 / 64 / the default through reserved[2], hundreds of launches, the key of EVERY launch and the counts of every 50th
    against the oracle.  A reader that ran ahead of another tile's counts would produce a key with too small a count."""
    torch, dev, ctx = gpu_ab
    n = 16384
    scene = synth.two_view_scene(n, seed=77)
    pair, _ = make_pair(S, gpu_ab, scene)
    p = S.default_params(n, num_hypotheses=H, seed=11, kernel=S.KERNEL_PREFILTER)
    p.reserved[2] = cols
    p.reserved[3] = var
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    key, ocounts, _ = O.ransac_range_fast(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed)
    bad_keys = 0
    for it in range(launches):
        pair.ransac_score(p)
        k = pair.get_key()
        if k != key:
            bad_keys += 1
        if it % 50 == 0:
            assert pair.last_launch()["kernel"] == S.KERNEL_PREFILTER
            assert np.array_equal(pair.get_inlier_counts(H), ocounts), f"launch {it}: counts differ"
    assert bad_keys == 0, f"{bad_keys} of {launches} launches produced a key other than the oracle's"


@pytest.mark.parametrize("mode", [2])
def test_prefilter_static_pass_order_equals_oracle(gpu_ab, mode):
    """reserved[1] = 2 hands the passes out by position instead of through the block's LDS counter (A/B switch): same counts."""
    n, H = 3000, 40000
    scene = synth.two_view_scene(n, seed=5)
    pair, _ = make_pair(S, gpu_ab, scene)
    p = S.default_params(n, num_hypotheses=H, seed=3, kernel=S.KERNEL_PREFILTER)
    p.reserved[1] = mode
    pair.estimateE(p)
    P.check_all(pair, scene, p, H, n)


def test_round2_kernel_still_equals_oracle(gpu_ab):
    """The round-2 scoring kernel kept for A/B runs (reserved[3] = 2)."""
    n, H = 4096, 65536
    scene = synth.two_view_scene(n, seed=8)
    pair, _ = make_pair(S, gpu_ab, scene)
    p = S.default_params(n, num_hypotheses=H, seed=2, kernel=S.KERNEL_PREFILTER)
    p.reserved[3] = 2
    pair.estimateE(p)
    assert pair.last_launch()["kernel"] == S.KERNEL_PREFILTER
    P.check_all(pair, scene, p, H, n)



@pytest.mark.parametrize("n,H,thr", [(4096, 65536, 1e-6), (1000, 20000, 1e-4), (16384, 16384, 1e-6), (700, 16385, 1e-8)])
def test_g_rule_kernel_still_equals_oracle(gpu_ab, n, H, thr):
    """The G rule of rounds 2-4 (a per-pair threshold contraction: v_fma + v_alignbit per pair, three MFMAs per 32 x 32 pairs), kept
    behind reserved[3] = 4 for A/B runs against the band rule: every count, key, E, mask -- and both rules' LDS footprints."""
    scene = synth.two_view_scene(n, seed=8)
    pair, _ = make_pair(S, gpu_ab, scene)
    p = S.default_params(n, num_hypotheses=H, seed=2, kernel=S.KERNEL_PREFILTER, threshold=thr)
    p.reserved[3] = 4
    pair.estimateE(p)
    g = pair.last_launch()
    assert g["kernel"] == S.KERNEL_PREFILTER
    P.check_all(pair, scene, p, H, n)
    q = S.default_params(n, num_hypotheses=H, seed=2, kernel=S.KERNEL_PREFILTER, threshold=thr)
    pair.estimateE(q)
    P.check_all(pair, scene, q, H, n)
    b = pair.last_launch()
    if n % 1024 == 0:
        assert b["lds_bytes"] + 32 * 1024 == g["lds_bytes"]              # 64 instead of 96 bytes of fragments per point of a 1024-point tile
    q.reserved[1] = 7                                                    # tiles of up to 1536 points (measured slower: profiles/r05_ab_tile_size.txt)
    pair.estimateE(q)
    assert pair.last_launch()["lds_bytes"] >= b["lds_bytes"]
    P.check_all(pair, scene, q, H, n)
    p.reserved[3] = 3                                                    # stand-alone record kernel, band rule
    pair.estimateE(p)
    P.check_all(pair, scene, p, H, n)


def test_polled_merge_give_up_is_reported_and_contained(gpu_ab):
    """The exact matcher's polled merge (match.hip, POLL) bounds its wait; a block that gives up must not leave silently wrong matches
    or a poisoned workspace behind.  SFM_MATCH_POLL_LIMIT=0 (lab-bench library only, read per call) makes the poller give up whenever a
    partial is not there at its first look: the give-up is reported once -- by the next sfm_match* call or sfm_ctx_synchronize, as
    SFM_E_HIP -- the queries concerned carry index -1 / score 0 (never a stale or half-merged value), and with the limit back the
    same context matches correctly again, through the polled merge and through the ticket-scheme fused matcher that shares the
    ticket workspace."""
    import os
    torch, dev, ctx = gpu_ab
    n1, n2 = 2155, 2112                # 14 splits of 160 database rows, the last one of 32: its block is done (and polls) long before the others
    d1, _, _ = synth.descriptors(n1, seed=31)
    d2, _, _ = synth.descriptors(n2, seed=32)
    ob, os_, oi = O.match_desc(d1, d2)
    t1, t2 = to_dev(torch, dev, d1), to_dev(torch, dev, d2)
    best = torch.empty(n1, dtype=torch.float32, device=dev); sec = torch.empty_like(best); idx = torch.empty(n1, dtype=torch.int32, device=dev)
    ctx.set_match_kernel(S.MATCH_EXACT)
    reported = 0
    try:
        os.environ["SFM_MATCH_POLL_LIMIT"] = "0"
        for it in range(200):
            best.fill_(-1.0); sec.fill_(-1.0); idx.fill_(-9)
            torch.cuda.synchronize()
            try:
                ctx.match_soa(t1, n1, 128, t2, n2, 128, best, sec, idx)
                ctx.synchronize()
            except S.SfmError as e:
                assert e.code == S.E_HIP and "polled merge" in str(e)
                reported += 1
                torch.cuda.synchronize()
                continue
            # no report: every query either matched exactly or -- if the give-up of THIS launch is still to be reported -- says "no match"
            i = idx.cpu().numpy(); b = best.cpu().numpy()
            gave = i == -1
            assert np.array_equal(i[~gave], oi[~gave]) and same_bits(b[~gave], ob[~gave]) and not (b[gave] != 0).any(), it
            if gave.any():
                with pytest.raises(S.SfmError):
                    ctx.synchronize()                                  # ... which the next synchronising call does
                reported += 1
            if reported >= 3:
                break
        assert reported >= 1, "SFM_MATCH_POLL_LIMIT=0 never made a poller give up in 200 launches"
    finally:
        os.environ.pop("SFM_MATCH_POLL_LIMIT", None)
    ctx.synchronize()                                                  # the flag was cleared by the report
    try:
        for kern in (S.MATCH_EXACT, S.MATCH_FUSED, S.MATCH_EXACT):
            ctx.set_match_kernel(kern)
            best.fill_(-1.0); sec.fill_(-1.0); idx.fill_(-9)
            ctx.match_soa(t1, n1, 128, t2, n2, 128, best, sec, idx)
            ctx.synchronize()
            assert np.array_equal(idx.cpu().numpy(), oi) and same_bits(best.cpu().numpy(), ob) and same_bits(sec.cpu().numpy(), os_), kern
    finally:
        ctx.set_match_kernel(S.MATCH_AUTO)


@pytest.mark.parametrize("n,H,thr", [(4096, 65536, 1e-6), (1000, 20000, 1e-4), (16384, 32768, 1e-6), (700, 16385, 1e-8), (5000, 20000, 1e-5)])
def test_per_hypothesis_records_still_equal_oracle(gpu_ab, n, H, thr):
    """The packed scan with per-hypothesis 64-byte records and whole-view boxes (reserved[3] = 6), kept for A/B runs against the
    product's per-tile band constants (tiles = runs of a Morton-ordered copy of the correspondences, sigma and the coefficient slots
    derived per (hypothesis, tile) inside the scoring kernel from a 4-byte flag record): every count, key, E, mask, either rule on the
    same pair -- and again after a second fillXU of other points (the ordered copy is rebuilt per fillXU epoch)."""
    scene = synth.two_view_scene(n, seed=12)
    pair, _ = make_pair(S, gpu_ab, scene)
    p = S.default_params(n, num_hypotheses=H, seed=6, kernel=S.KERNEL_PREFILTER, threshold=thr)
    p.reserved[3] = 6
    pair.estimateE(p)
    assert pair.last_launch()["kernel"] == S.KERNEL_PREFILTER
    P.check_all(pair, scene, p, H, n)
    q = S.default_params(n, num_hypotheses=H, seed=6, kernel=S.KERNEL_PREFILTER, threshold=thr)      # the product's rule on the same pair
    pair.estimateE(q)
    P.check_all(pair, scene, q, H, n)
    torch, dev, ctx = gpu_ab
    scene2 = synth.two_view_scene(n, seed=13)
    pair.fillXU(to_dev(torch, dev, scene2["sift"]))
    pair.estimateE(q)
    P.check_all(pair, scene2, q, H, n)
    pair.estimateE(p)
    P.check_all(pair, scene2, p, H, n)


@pytest.mark.parametrize("n,H", [(4096, 32768), (3000, 20000), (900, 16385)])
def test_wide_ring_entries_still_equal_oracle(gpu_ab, n, H):
    """Recorded variant (reserved[1] = 13): ring entries of 16 bytes covering four 32-point steps (two conversions) instead of 8 bytes
    per two steps -- half as many appends, measured 3 % slower (profiles/r06_ab_wide_entries.txt): every count, key, E, mask."""
    scene = synth.two_view_scene(n, seed=14)
    pair, _ = make_pair(S, gpu_ab, scene)
    p = S.default_params(n, num_hypotheses=H, seed=8, kernel=S.KERNEL_PREFILTER)
    p.reserved[1] = 13
    pair.estimateE(p)
    assert pair.last_launch()["kernel"] == S.KERNEL_PREFILTER
    P.check_all(pair, scene, p, H, n)


def _view_box_case():
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prefilter_view_box_case.npz"))
    n = int(d["n"])
    sift = np.zeros(n, synth.SIFT_DTYPE)
    for f in ("xpos", "ypos", "match_xpos", "match_ypos"):
        sift[f] = d[f]
    return {"sift": sift, "K": d["K"], "Kinv": d["Kinv"]}, n, int(d["H"]), float(d["thr"]), int(d["seed"])


@pytest.mark.parametrize("sweeps", [0, 7])
def test_view_boxes_of_every_record_builder_on_the_case_the_fuzz_found(gpu, gpu_ab, sweeps):
    """Round 6: hipcc dropped the negation of ONE lower bound of the whole-view boxes in the stand-alone record kernel (pf_box_from_words:
    xlo = +max(-x), a box smaller than the points' range, sigma up to 20 % too large) -- 23 (Jacobi solver) / 152 (Householder + stand-alone
    records) of 1607 hypotheses of this scene lost inliers; the lane-solve kernel's copy of the same function was right, so only paths behind
    pf_prep_kernel showed it (profiles/NOTES_r06.md).  Every per-hypothesis record builder and the per-tile form, both solvers, both calls."""
    scene, n, H, thr, seed = _view_box_case()
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    key, ocounts, _ = O.ransac_range(X0, X1, 0, H, np.float32(thr), sweeps, seed=seed)
    for lib, g, r3 in ((PROD, gpu, 0), (S, gpu_ab, 6), (S, gpu_ab, 7), (S, gpu_ab, 5), (S, gpu_ab, 3)):
        pair, _ = make_pair(lib, g, scene)
        p = lib.default_params(n, num_hypotheses=H, seed=seed, kernel=lib.KERNEL_PREFILTER, jacobi_sweeps=sweeps, threshold=thr)
        p.reserved[3] = r3
        for call in range(2):
            pair.estimateE(p)
            assert pair.last_launch()["kernel"] == lib.KERNEL_PREFILTER
            c = pair.get_inlier_counts(H)
            bad = np.flatnonzero(c != ocounts)
            assert bad.size == 0, f"reserved[3] = {r3}, call {call + 1}: {bad.size} counts differ, gpu - oracle in [{int((c - ocounts).min())}, {int((c - ocounts).max())}]"
            assert pair.get_key() == key
        pair.close()
