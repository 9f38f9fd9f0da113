"""GPU parity: MFMA descriptor matcher against the oracle (bit-exact scores and indices)."""
import numpy as np
import pytest

import cuda_sfm_amd as S
from cuda_sfm_amd import synth
import oracle as O
from helpers import same_bits, to_dev

pytestmark = pytest.mark.gpu


def run_soa(gpu, d1, d2):
    torch, dev, ctx = gpu
    t1, t2 = to_dev(torch, dev, d1), to_dev(torch, dev, d2)
    n1, n2 = d1.shape[0], d2.shape[0]
    best = torch.empty(n1, dtype=torch.float32, device=dev)
    sec = torch.empty(n1, dtype=torch.float32, device=dev)
    idx = torch.empty(n1, dtype=torch.int32, device=dev)
    ctx.match_soa(t1, n1, d1.shape[1], t2, n2, d2.shape[1], best, sec, idx)
    torch.cuda.synchronize()
    return best.cpu().numpy(), sec.cpu().numpy(), idx.cpu().numpy()


@pytest.mark.parametrize("n1,n2", [(1, 1), (5, 3), (31, 33), (64, 64), (100, 777), (512, 512), (1911, 2086), (4100, 1000)])
def test_match_soa_bit_exact(gpu, n1, n2):
    d1, _, _ = synth.descriptors(n1, seed=100 + n1)
    d2, _, _ = synth.descriptors(n2, seed=200 + n2)
    b, s, i = run_soa(gpu, d1, d2)
    ob, os_, oi = O.match_desc(d1, d2)
    assert np.array_equal(i, oi)
    assert same_bits(b, ob) and same_bits(s, os_)


@pytest.mark.parametrize("n1,n2", [(2155, 2170), (1500, 1500), (130, 1900), (2400, 96), (700, 2530), (3000, 161), (257, 33)])
def test_exact_matcher_split_tails(gpu, n1, n2):
    """The exact MFMA matcher with database splits that are an odd number of 32-row tiles (the last stage of a split then runs one
    row tile, match.hip: stage_scores<CT, 1>), with more than sixteen splits (two merge trips) and with a single short split."""
    import cuda_sfm_amd as S
    torch, dev, ctx = gpu
    d1, _, _ = synth.descriptors(n1, seed=300 + n1)
    d2, _, _ = synth.descriptors(n2, seed=400 + n2)
    ctx.set_match_kernel(S.MATCH_EXACT)
    try:
        b, s, i = run_soa(gpu, d1, d2)
    finally:
        ctx.set_match_kernel(S.MATCH_AUTO)
    ob, os_, oi = O.match_desc(d1, d2)
    assert np.array_equal(i, oi)
    assert same_bits(b, ob) and same_bits(s, os_)


def test_match_recovers_permutation_and_ties(gpu):
    n = 2048
    d1, d2, perm = synth.descriptors(n)
    b, s, i = run_soa(gpu, d2, d1)            # query = set 2, database = set 1 -> index = perm
    assert (i == perm).mean() > 0.99
    # exact duplicates in the database: lowest index must win, second == best
    d2b = d1.copy(); d2b[1000] = d2b[10]
    b, s, i = run_soa(gpu, d1[10:11], d2b)
    assert i[0] == 10 and b[0] == s[0]
    ob, os_, oi = O.match_desc(d1[10:11], d2b)
    assert oi[0] == 10 and same_bits(b, ob) and same_bits(s, os_)


def test_match_no_positive_score(gpu):
    """Scores <= 0 never win (scores start at 0, strict '>', match.cu:59-68): index stays -1."""
    d1 = np.zeros((3, 128), np.float32); d1[:, 0] = 1.0
    d2 = np.zeros((70, 128), np.float32); d2[:, 0] = -1.0
    b, s, i = run_soa(gpu, d1, d2)
    assert (i == -1).all() and (b == 0).all() and (s == 0).all()


def test_match_sift_records(gpu):
    """MatchSiftData semantics on SiftPoint AoS: in-place score/match/match_xpos/match_ypos/ambiguity."""
    torch, dev, ctx = gpu
    n1, n2 = 700, 900
    d1, _, _ = synth.descriptors(n1, seed=1)
    d2, _, _ = synth.descriptors(n2, seed=2)
    s1 = synth.sift_records(d1, seed=3); s2 = synth.sift_records(d2, seed=4)
    t1, t2 = to_dev(torch, dev, s1), to_dev(torch, dev, s2)
    ctx.match(t1, n1, t2, n2)
    torch.cuda.synchronize()
    out = t1.cpu().numpy().reshape(-1).view(synth.SIFT_DTYPE)
    ref = O.match_sift(s1, s2)
    for f in ("score", "ambiguity", "match_xpos", "match_ypos"):
        assert same_bits(out[f], ref[f]), f
    assert np.array_equal(out["match"], ref["match"])
    for f in ("xpos", "ypos", "scale", "data"):
        assert np.array_equal(out[f], s1[f])          # untouched fields
    ctx.match(t1, 0, t2, n2)                          # matching.cu:1095-1096 early-out


def test_match_full_size_properties(gpu):
    """16384 x 16384 (match.cu benchmark size; the north_star size of the matcher): permutation recovery, and EVERY query's
    (best, second, index) against the CPU restatement of MatchC1 (CudaSift/match.cu:57-71) bit for bit -- a sweep, not a sample:
    the oracle matcher needs ~6 s for the 2.7e8 pairs."""
    n = 16384
    d1, d2, perm = synth.descriptors(n)
    b, s, i = run_soa(gpu, d2, d1)
    assert (i == perm).mean() > 0.99
    assert (b >= s).all() and (i >= 0).all()
    ob, os_, oi = O.match_desc(d2, d1)
    bad = np.flatnonzero(i != oi)
    assert bad.size == 0, f"{bad.size} of {n} indices differ, first queries {bad[:5]}"
    assert same_bits(b, ob) and same_bits(s, os_)


def test_exact_polled_merge_next_to_the_other_matchers(gpu):
    """The polled merge of the exact matcher trusts a partial by its epoch tag alone, so nothing else may ever write where it polls:
    the fused matcher's ticket workspace (floats and row indices: an index equal to the launch counter in the upper half of a polled
    word would pass for a fresh partial) is a different buffer.  150 alternations of fused and exact launches on one context, sizes
    whose indices sweep the range of the epochs, every result against the oracle's."""
    torch, dev, ctx = gpu
    rng = np.random.default_rng(5)
    n1, n2 = 1500, 2049
    d1, _, _ = synth.descriptors(n1, seed=11)
    d2, _, _ = synth.descriptors(n2, seed=12)
    ob, os_, oi = O.match_desc(d1, d2)
    t1, t2 = to_dev(torch, dev, d1), to_dev(torch, dev, d2)
    best = torch.empty(n1, dtype=torch.float32, device=dev); sec = torch.empty_like(best); idx = torch.empty(n1, dtype=torch.int32, device=dev)
    try:
        for it in range(150):
            ctx.set_match_kernel(S.MATCH_FUSED)
            ctx.match_soa(t1, n1, 128, t2, n2, 128, best, sec, idx)
            ctx.set_match_kernel(S.MATCH_EXACT)
            best.fill_(-1.0); sec.fill_(-1.0); idx.fill_(-9)
            ctx.match_soa(t1, n1, 128, t2, n2, 128, best, sec, idx)
            torch.cuda.synchronize()
            assert ctx.last_match_kernel() == S.MATCH_EXACT
            assert np.array_equal(idx.cpu().numpy(), oi) and same_bits(best.cpu().numpy(), ob) and same_bits(sec.cpu().numpy(), os_), it
    finally:
        ctx.set_match_kernel(S.MATCH_AUTO)
