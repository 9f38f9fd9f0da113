"""GPU parity: the fp16 matrix-core pre-filter matchers (match_prefilter.hip: four launches, global threshold; match_fused.hip:
one launch, running threshold) against the exact MFMA matcher and the oracle.  Bit-exact scores and indices are the bar:
the pre-filters only select which rows get the exact fp32 chain."""
import numpy as np
import pytest

import cuda_sfm_amd as S
from cuda_sfm_amd import synth
import oracle as O
from helpers import same_bits, to_dev

pytestmark = pytest.mark.gpu


def run_soa(gpu, d1, d2, kernel):
    torch, dev, ctx = gpu
    ctx.set_match_kernel(kernel)
    try:
        t1, t2 = to_dev(torch, dev, d1), to_dev(torch, dev, d2)
        n1, n2 = d1.shape[0], d2.shape[0]
        best = torch.full((n1,), -5.0, dtype=torch.float32, device=dev)
        sec = torch.full((n1,), -5.0, dtype=torch.float32, device=dev)
        idx = torch.full((n1,), -7, dtype=torch.int32, device=dev)
        ctx.match_soa(t1, n1, d1.shape[1], t2, n2, d2.shape[1], best, sec, idx)
        torch.cuda.synchronize()
        ran = ctx.last_match_kernel()
    finally:
        ctx.set_match_kernel(S.MATCH_AUTO)
    return best.cpu().numpy(), sec.cpu().numpy(), idx.cpu().numpy(), ran


def check_both(gpu, d1, d2, oracle_rows=None):
    b, s, i, ran = run_soa(gpu, d1, d2, S.MATCH_PREFILTER)
    assert ran == S.MATCH_PREFILTER
    eb, es, ei, ran2 = run_soa(gpu, d1, d2, S.MATCH_EXACT)
    assert ran2 == S.MATCH_EXACT
    assert np.array_equal(i, ei)
    assert same_bits(b, eb) and same_bits(s, es)
    fb, fs, fi, ran3 = run_soa(gpu, d1, d2, S.MATCH_FUSED)
    assert ran3 == S.MATCH_FUSED
    bad = np.flatnonzero(fi != ei)
    assert bad.size == 0, f"fused: {bad.size} indices differ, first queries {bad[:5]}: {fi[bad[:5]]} vs {ei[bad[:5]]}"
    assert same_bits(fb, eb) and same_bits(fs, es)
    rows = np.arange(d1.shape[0]) if oracle_rows is None else oracle_rows
    ob, os_, oi = O.match_desc(d1[rows], d2)
    assert np.array_equal(i[rows], oi) and same_bits(b[rows], ob) and same_bits(s[rows], os_)
    return b, s, i


@pytest.mark.parametrize("n1,n2", [(1, 1), (5, 3), (31, 33), (64, 64), (100, 777), (513, 511), (1911, 2086), (3000, 3100), (4100, 1000)])
def test_prefilter_bit_exact(gpu, n1, n2):
    d1, _, _ = synth.descriptors(n1, seed=300 + n1)
    d2, _, _ = synth.descriptors(n2, seed=400 + n2)
    check_both(gpu, d1, d2)


def test_prefilter_permutation_and_duplicates(gpu):
    n = 4096
    d1, d2, perm = synth.descriptors(n)
    b, s, i = check_both(gpu, d2, d1, oracle_rows=np.arange(0, n, 37))
    assert (i == perm).mean() > 0.99
    # exact duplicates in the database: lowest index wins, second == best
    db = d1.copy(); db[3000] = db[10]; db[77] = db[10]
    b, s, i = check_both(gpu, d1[:64], db)
    assert i[10] == 10 and b[10] == s[10]


def test_prefilter_many_equal_rows_take_the_full_scan(gpu):
    """More candidates than slots (every database row identical, or one row repeated 500 times): the exact kernel scans all rows."""
    rng = np.random.default_rng(5)
    d1, _, _ = synth.descriptors(300, seed=9)
    row = d1[7].copy()
    db = np.tile(row, (2000, 1)).astype(np.float32)
    b, s, i = check_both(gpu, d1, db)
    assert (i == 0).all() and same_bits(b, s)
    db2, _, _ = synth.descriptors(2500, seed=10)
    db2[rng.choice(2500, 500, replace=False)] = d1[3]
    check_both(gpu, d1, db2)


def test_prefilter_signs_zeros_tiny_and_scaled(gpu):
    rng = np.random.default_rng(11)
    n1, n2 = 257, 1500
    d1 = rng.standard_normal((n1, 128)).astype(np.float32)
    d2 = rng.standard_normal((n2, 128)).astype(np.float32)
    check_both(gpu, d1, d2)                                    # mixed signs, norms ~ 11
    d1z = d1.copy(); d1z[5] = 0.0; d1z[6, 1:] = 0.0
    d2z = d2.copy(); d2z[100] = 0.0
    check_both(gpu, d1z, d2z)                                  # zero rows: scores 0 never win
    check_both(gpu, (d1 * 1e-6).astype(np.float32), (d2 * 1e-3).astype(np.float32))    # below the fp16 floor of the scaled copies
    check_both(gpu, (d1 * 20.0).astype(np.float32), (d2 * 20.0).astype(np.float32))    # entries up to ~90: still inside the fp16 range
    neg = -np.abs(d2)
    b, s, i = check_both(gpu, np.abs(d1), neg)                 # no positive score at all
    assert (i == -1).all() and (b == 0).all() and (s == 0).all()


def test_prefilter_entries_the_fp16_copy_cannot_hold(gpu):
    """|entry| > 255, inf, NaN in either set: those queries (or all of them, for a database row) take the full exact scan."""
    rng = np.random.default_rng(12)
    d1, _, _ = synth.descriptors(200, seed=21)
    d2, _, _ = synth.descriptors(1200, seed=22)
    q = d1.copy(); q[3, 5] = 1000.0; q[9, 0] = 3e38
    check_both(gpu, q, d2)
    db = d2.copy(); db[50, 7] = 400.0
    check_both(gpu, d1, db)
    db = d2.copy(); db[60] *= 1e4
    check_both(gpu, d1, db)
    # the offending entry anywhere in the row (the row's sixteen lanes must all learn of it), several rows at once
    q = d1.copy(); q[4, 77] = 1000.0; q[5, 127] = -2000.0; q[6, 8] = 300.0
    db = d2.copy(); db[70, 127] = 500.0; db[71, 64] = -256.5
    check_both(gpu, q, d2)
    check_both(gpu, d1, db)
    check_both(gpu, q, db)


def test_prefilter_sift_records_and_auto(gpu):
    """MatchSiftData semantics through the pre-filter paths; AUTO: exact, then fused, the four-kernel pre-filter from 6144 x 6144 on."""
    torch, dev, ctx = gpu
    n1, n2 = 6200, 6150
    d1, _, _ = synth.descriptors(n1, seed=31)
    d2, _, _ = synth.descriptors(n2, seed=32)
    s1 = synth.sift_records(d1, seed=33); s2 = synth.sift_records(d2, seed=34)
    t1, t2 = to_dev(torch, dev, s1), to_dev(torch, dev, s2)
    ctx.set_match_kernel(S.MATCH_AUTO)
    ctx.match(t1, n1, t2, n2)
    torch.cuda.synchronize()
    assert ctx.last_match_kernel() == S.MATCH_PREFILTER
    out = t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)
    ref = O.match_sift(s1, s2)
    for f in ("score", "ambiguity", "match_xpos", "match_ypos"):
        assert same_bits(out[f], ref[f]), f
    assert np.array_equal(out["match"], ref["match"])
    for f in ("xpos", "ypos", "scale", "data"):
        assert np.array_equal(out[f], s1[f])
    ctx.match(t1, 700, t2, 900)
    assert ctx.last_match_kernel() == S.MATCH_EXACT
    ctx.match(t1, 3000, t2, 3100)
    assert ctx.last_match_kernel() == S.MATCH_FUSED
    torch.cuda.synchronize()
    out = t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)[:3000]
    ref = O.match_sift(s1[:3000].copy(), s2[:3100])
    for f in ("score", "ambiguity", "match_xpos", "match_ypos"):
        assert same_bits(out[f], ref[f]), f
    assert np.array_equal(out["match"], ref["match"])


def test_auto_on_plain_descriptor_arrays(gpu):
    """sfm_match_soa with rows 512 bytes apart: AUTO keeps the exact matcher up to 3400^2 and takes the four-kernel pre-filter
    above (the fused matcher's scattered loads dislike that stride); results as always."""
    d1, d2, perm = synth.descriptors(4096)
    b, s, i, ran = run_soa(gpu, d2, d1, S.MATCH_AUTO)
    assert ran == S.MATCH_PREFILTER and (i == perm).mean() > 0.99
    b, s, i, ran = run_soa(gpu, d2[:3000], d1[:3000], S.MATCH_AUTO)
    assert ran == S.MATCH_EXACT
    ob, os_, oi = O.match_desc(d2[:50], d1[:3000])
    assert np.array_equal(i[:50], oi) and same_bits(b[:50], ob) and same_bits(s[:50], os_)


def test_fused_many_stages_duplicates_across_stages(gpu):
    """Enough query blocks that a block walks several 128-row stages of the database: duplicated rows in DIFFERENT stages
    (the lowest index wins, second == best), a query whose best comes late, equal rows filling a whole stage."""
    rng = np.random.default_rng(41)
    n1, n2 = 8192, 2500
    base, _, _ = synth.descriptors(n2, seed=51)
    q = base[rng.integers(0, n2, n1)].copy()
    q += rng.normal(scale=0.002, size=q.shape).astype(np.float32)
    db = base.copy()
    db[2000] = db[10]; db[1300] = db[10]; db[140] = db[139]
    db[1500:1700] = db[77]                                     # more than a stage of equal rows
    check_both(gpu, q, db, oracle_rows=np.arange(0, n1, 61))


def test_prefilter_full_size(gpu):
    """16384 x 16384 (match.cu benchmark size, the size north_star names for the matcher): EXACT, PREFILTER and FUSED agree on
    every query, and all 16384 queries (best, second, index) equal the CPU oracle's MatchC1 restatement bit for bit -- the fp16
    pre-filters are checked against the oracle on the whole set, not against the HIP path itself."""
    n = 16384
    d1, d2, perm = synth.descriptors(n)
    b, s, i = check_both(gpu, d2, d1)                          # oracle_rows=None: every row
    assert (i == perm).mean() > 0.99


def test_fused_random_shapes_and_data(gpu):
    """Forty random shapes and data families through the fused matcher against the exact one (every query, bit for bit):
    ragged sizes around the 128-row stage and the 128-query block, clustered descriptors (many near-equal scores: long
    candidate lists), duplicated rows across stages, sparse non-negative rows, mixed signs."""
    rng = np.random.default_rng(2026)
    for trial in range(40):
        n1 = int(rng.choice([1, 31, 127, 128, 129, 300, 700, 1500, 2600]))
        n2 = int(rng.choice([1, 2, 33, 127, 128, 129, 255, 257, 900, 2049, 3500]))
        family = trial % 5
        if family == 0:
            d1 = np.abs(rng.standard_normal((n1, 128))).astype(np.float32); d2 = np.abs(rng.standard_normal((n2, 128))).astype(np.float32)
        elif family == 1:                                   # clusters: rows are small perturbations of a few centres
            centres = np.abs(rng.standard_normal((6, 128))).astype(np.float32)
            d1 = centres[rng.integers(0, 6, n1)] + 1e-3 * rng.standard_normal((n1, 128)).astype(np.float32)
            d2 = centres[rng.integers(0, 6, n2)] + 1e-3 * rng.standard_normal((n2, 128)).astype(np.float32)
        elif family == 2:                                   # exact duplicates spread over the database
            base = np.abs(rng.standard_normal((max(2, n2 // 7), 128))).astype(np.float32)
            d2 = base[rng.integers(0, base.shape[0], n2)].copy()
            d1 = base[rng.integers(0, base.shape[0], n1)].copy()
        elif family == 3:                                   # sparse
            d1 = (np.abs(rng.standard_normal((n1, 128))) * (rng.random((n1, 128)) < 0.15)).astype(np.float32)
            d2 = (np.abs(rng.standard_normal((n2, 128))) * (rng.random((n2, 128)) < 0.15)).astype(np.float32)
        else:
            d1 = rng.standard_normal((n1, 128)).astype(np.float32); d2 = rng.standard_normal((n2, 128)).astype(np.float32)
        if family != 4:
            d1 /= np.maximum(np.linalg.norm(d1, axis=1, keepdims=True), 1e-20); d2 /= np.maximum(np.linalg.norm(d2, axis=1, keepdims=True), 1e-20)
        fb, fs, fi, ran = run_soa(gpu, d1, d2, S.MATCH_FUSED)
        assert ran == S.MATCH_FUSED
        eb, es, ei, _ = run_soa(gpu, d1, d2, S.MATCH_EXACT)
        bad = np.flatnonzero((fi != ei) | (fb.view(np.uint32) != eb.view(np.uint32)) | (fs.view(np.uint32) != es.view(np.uint32)))
        assert bad.size == 0, f"trial {trial} ({n1} x {n2}, family {family}): {bad.size} queries differ, first {bad[:5]}"
        if trial % 8 == 0:
            rows = np.arange(0, n1, max(1, n1 // 50))
            ob, os_, oi = O.match_desc(d1[rows], d2)
            assert np.array_equal(fi[rows], oi) and same_bits(fb[rows], ob) and same_bits(fs[rows], os_)


def test_fused_many_matches_launch_equals_single_matches(gpu):
    """The many-matches launch of sfm_process_pairs (grid.z = match, one split per match when there are enough blocks) against
    one sfm_match call per pair: index arrays and the record fields of every first view."""
    torch, dev, ctx = gpu
    V, n = 7, 1337
    K, Kinv = synth.camera()
    rng = np.random.default_rng(77)
    centres = np.abs(rng.standard_normal((40, 128))).astype(np.float32)
    views = []
    for v in range(V):
        d = centres[rng.integers(0, 40, n)] + 0.05 * np.abs(rng.standard_normal((n, 128))).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        views.append(synth.sift_records(d.astype(np.float32), seed=500 + v))
    pairs_host = [(i, j) for i in range(V) for j in range(i + 1, V)]
    want = {}
    for (i, j) in pairs_host:                              # sequential: the fields of view i after its LAST pair are what remain
        t1, t2 = to_dev(torch, dev, views[i]), to_dev(torch, dev, views[j])
        ctx.match(t1, n, t2, n)
        torch.cuda.synchronize()
        want[i] = t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE).copy()
    dviews = [to_dev(torch, dev, v) for v in views]
    descs = [(dviews[i], n, dviews[j], n) for (i, j) in pairs_host]
    rec, status = S.process_pairs_local(ctx, descs, K, Kinv)
    torch.cuda.synchronize()
    for i in range(V - 1):
        got = dviews[i].cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)
        assert np.array_equal(got["match"], want[i]["match"]), f"first view {i}"
        for f in ("score", "ambiguity", "match_xpos", "match_ypos"):
            assert same_bits(got[f], want[i][f]), (i, f)
