"""Parity soak of the pre-filter matcher: random sizes and descriptor flavours (SIFT-like, clustered around a few centres so
that many scores are nearly equal, exact duplicates, signed, tiny / large scales, entries beyond the fp16 range), every
query's (best, second, index) of SFM_MATCH_PREFILTER against SFM_MATCH_EXACT bit for bit (the test-suite pins the exact
matcher to the oracle).    python profiles/match_soak.py [seconds] [seed]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)


def sift_like(n):
    x = np.abs(rng.standard_normal((n, 128))) ** 3
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x = np.minimum(x, 0.2)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def make(flavour, n1, n2):
    if flavour == "sift":
        return sift_like(n1), sift_like(n2)
    if flavour == "clustered":                      # few centres + small noise: many rows score within eps of each other
        c = sift_like(int(rng.integers(2, 40)))
        noise = float(10.0 ** rng.uniform(-6, -2))
        q = c[rng.integers(0, len(c), n1)] + noise * rng.standard_normal((n1, 128))
        d = c[rng.integers(0, len(c), n2)] + noise * rng.standard_normal((n2, 128))
        return q.astype(np.float32), d.astype(np.float32)
    if flavour == "dup":
        q, d = sift_like(n1), sift_like(n2)
        k = max(1, n2 // 5)
        d[rng.integers(0, n2, k)] = d[rng.integers(0, n2, k)]
        d[rng.integers(0, n2, min(n1, k))] = q[rng.integers(0, n1, min(n1, k))]
        return q, d
    if flavour == "signed":
        s = float(10.0 ** rng.uniform(-7, 1.3))
        return (rng.standard_normal((n1, 128)) * s).astype(np.float32), (rng.standard_normal((n2, 128)) * s).astype(np.float32)
    q, d = sift_like(n1), sift_like(n2)             # "wild": a few entries the fp16 copy cannot hold, zero rows
    for a in (q, d):
        for _ in range(int(rng.integers(0, 4))):
            a[rng.integers(0, a.shape[0]), rng.integers(0, 128)] = float(rng.choice([300.0, -1e4, 3e30, 256.0]))
        if rng.random() < 0.5:
            a[rng.integers(0, a.shape[0])] = 0.0
    return q, d


def run(kernel, t1, n1, t2, n2):
    best = torch.full((n1,), -5.0, dtype=torch.float32, device=dev); sec = torch.full((n1,), -5.0, dtype=torch.float32, device=dev)
    idx = torch.full((n1,), -7, dtype=torch.int32, device=dev)
    ctx.set_match_kernel(kernel)
    ctx.match_soa(t1, n1, 128, t2, n2, 128, best, sec, idx)
    torch.cuda.synchronize()
    return best.cpu().numpy().view(np.uint32), sec.cpu().numpy().view(np.uint32), idx.cpu().numpy()


t_end = time.time() + budget
rounds, queries, bad = 0, 0, []
per = {}
while time.time() < t_end:
    flavour = str(rng.choice(["sift", "clustered", "dup", "signed", "wild"]))
    n1 = int(rng.choice([1, 7, 33, 100, 511, 513, 1000, 2500, 4097, 6000]) if rng.random() < 0.5 else rng.integers(1, 6000))
    n2 = int(rng.choice([1, 5, 64, 127, 129, 1000, 3000, 5000]) if rng.random() < 0.5 else rng.integers(1, 6000))
    q, d = make(flavour, n1, n2)
    with np.errstate(all="ignore"):
        t1, t2 = torch.from_numpy(q).to(dev), torch.from_numpy(d).to(dev)
    a = run(S.MATCH_EXACT, t1, n1, t2, n2)
    b = run(S.MATCH_PREFILTER, t1, n1, t2, n2)
    ok = all(np.array_equal(x, y) for x, y in zip(a, b))
    rounds += 1; queries += n1; per[flavour] = per.get(flavour, 0) + 1
    if not ok:
        bad.append({"flavour": flavour, "n1": n1, "n2": n2, "differing_queries": int((a[2] != b[2]).sum()), "differing_best": int((a[0] != b[0]).sum()), "differing_second": int((a[1] != b[1]).sum())})
ctx.set_match_kernel(S.MATCH_AUTO)
print(json.dumps({"seconds": budget, "rounds": rounds, "per_flavour": per, "queries_checked": queries, "mismatches": bad[:20], "mismatch_count": len(bad)}))
