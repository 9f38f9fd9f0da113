"""Narrowing down pre-filter counts that differ from the plain kernel's (the cases of profiles/soak_cases_r02.json with the
draws profiles/soak_repro.py found): which point is lost, and does it depend on where the point sits in the tile?"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd_ab as S            # the lab-bench flavour (make ab): switches, probes, traces
from cuda_sfm_amd_ab import synth

dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
CASES = [dict(n=8192, focal=200.0, noise=1.5, of=0.8, seed=901921752, thr=0.00024245568783953786, hyp=229749),
         dict(n=12000, focal=2360.0, noise=0.0, of=0.5, seed=861219769, thr=2.8631911845877767e-05, hyp=133602),
         dict(n=12000, focal=200.0, noise=1.5, of=0.5, seed=316618012, thr=0.00037952669663354754, hyp=51262)]
H = 1 << 18
DONOR = {"xpos": 10.0, "ypos": 500.0, "match_xpos": 700.0, "match_ypos": 20.0}
for ci, c in enumerate(CASES):
    n, thr, hyp, seed = c["n"], float(np.float32(c["thr"])), c["hyp"], c["seed"]
    sc = synth.two_view_scene(n, seed=seed, focal=c["focal"], noise_px=c["noise"], outlier_frac=c["of"])
    sift = sc["sift"].copy()
    pair = S.ImagePair(ctx, sc["K"], sc["Kinv"], 2, n)

    def fill(s):
        pair.fillXU(torch.from_numpy(s.view(np.uint8).reshape(n, 576)).to(dev))

    def counts_for(E, kernel):
        k = E.shape[0]
        p = S.default_params(n, num_hypotheses=k, seed=1, kernel=kernel, threshold=thr, jacobi_sweeps=7)
        pair.ransac_score_candidates(p, torch.from_numpy(np.ascontiguousarray(E, np.float32)).to(dev))
        torch.cuda.synchronize()
        return pair.get_inlier_counts(k).copy()

    def only(points, at=None):
        """scene with the donor everywhere except the given points (optionally moved to other indices)"""
        s = sift.copy()
        for f in DONOR:
            v = np.full(n, np.float32(DONOR[f]))
            for k, j in enumerate(points):
                v[j if at is None else at[k]] = sift[f][j]
            s[f] = v
        return s

    fill(sift)
    p = S.default_params(n, num_hypotheses=H, seed=seed & 0xFFFF, kernel=S.KERNEL_SPLIT, threshold=thr, jacobi_sweeps=7)
    pair.estimateE(p)
    E1 = pair.get_E_candidates(H)[hyp:hyp + 1].copy()
    E64 = np.repeat(E1, 64, 0)
    a, b = counts_for(E64, S.KERNEL_SPLIT), counts_for(E64, S.KERNEL_PREFILTER)
    print("case", ci, "full scene: split", int(a[0]), "prefilter", int(b[0]), flush=True)
    lo, hi = 0, n
    while hi - lo > 1:
        mid = (lo + hi) // 2
        fill(only(range(lo, mid)))
        a, b = counts_for(E64, S.KERNEL_SPLIT), counts_for(E64, S.KERNEL_PREFILTER)
        if a[0] != b[0]:
            hi = mid
        else:
            lo = mid
    j = lo
    fill(only([j]))
    a, b = counts_for(E64, S.KERNEL_SPLIT), counts_for(E64, S.KERNEL_PREFILTER)
    X0, X1 = pair.get_XU(S.BUF_X0), pair.get_XU(S.BUF_X1)
    print("case", ci, "suspect point", j, "tile", j // 1024, "in-tile", j % 1024, "block", (j % 1024) // 32, "col", j % 32, "alone: split", int(a[0]), "prefilter", int(b[0]),
          "x1", X0[:2, j].tolist(), "x2", X1[:2, j].tolist(), "E", E1.ravel().tolist(), "thr", thr, flush=True)
    for at in (0, 1, 31, 32, 33, 63, 64, 278, 500, 1023, 1024, 1024 + 278, 5000):
        if at >= n:
            continue
        fill(only([j], [at]))
        a, b = counts_for(E64, S.KERNEL_SPLIT), counts_for(E64, S.KERNEL_PREFILTER)
        print("   moved to index", at, ": split", int(a[0]), "prefilter", int(b[0]), flush=True)
    # a different donor (changes the tile bound B and the occupied cells)
    for dn in ({"xpos": 360.0, "ypos": 288.0, "match_xpos": 300.0, "match_ypos": 200.0}, {"xpos": 100.0, "ypos": 100.0, "match_xpos": 600.0, "match_ypos": 500.0}):
        s = sift.copy()
        for f in dn:
            v = np.full(n, np.float32(dn[f])); v[j] = sift[f][j]; s[f] = v
        fill(s)
        a, b = counts_for(E64, S.KERNEL_SPLIT), counts_for(E64, S.KERNEL_PREFILTER)
        print("   donor", dn, ": split", int(a[0]), "prefilter", int(b[0]), flush=True)
    # thresholds around the case's
    fill(only([j]))
    for f in (0.5, 0.9, 0.99, 1.0, 1.01, 1.1, 2.0, 4.0):
        thr_keep = thr
        thr = float(np.float32(thr_keep * f))
        a, b = counts_for(E64, S.KERNEL_SPLIT), counts_for(E64, S.KERNEL_PREFILTER)
        print("   threshold x", f, ": split", int(a[0]), "prefilter", int(b[0]), flush=True)
        thr = thr_keep
    pair.close()
