"""Soak of the fused matcher against the exact one: random shapes / data families for SECONDS (default 120), every query bit for bit."""
import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cuda_sfm_amd as S
from helpers import to_dev
dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
t_end = time.time() + float(os.environ.get("SECONDS", "120"))
pairs = 0; trials = 0


def run(d1, d2, k):
    ctx.set_match_kernel(k)
    t1, t2 = to_dev(torch, dev, d1), to_dev(torch, dev, d2)
    n1, n2 = d1.shape[0], d2.shape[0]
    b = torch.empty(n1, dtype=torch.float32, device=dev); s = torch.empty(n1, dtype=torch.float32, device=dev); i = torch.empty(n1, dtype=torch.int32, device=dev)
    ctx.match_soa(t1, n1, 128, t2, n2, 128, b, s, i)
    torch.cuda.synchronize()
    return b.cpu().numpy().view(np.uint32), s.cpu().numpy().view(np.uint32), i.cpu().numpy()


while time.time() < t_end:
    n1 = int(rng.integers(1, 6000)); n2 = int(rng.integers(1, 9000))
    fam = int(rng.integers(0, 5))
    if fam == 0:
        d1 = np.abs(rng.standard_normal((n1, 128))); d2 = np.abs(rng.standard_normal((n2, 128)))
    elif fam == 1:
        c = np.abs(rng.standard_normal((int(rng.integers(2, 50)), 128)))
        d1 = c[rng.integers(0, len(c), n1)] + 10.0 ** rng.uniform(-6, -1) * rng.standard_normal((n1, 128))
        d2 = c[rng.integers(0, len(c), n2)] + 10.0 ** rng.uniform(-6, -1) * rng.standard_normal((n2, 128))
    elif fam == 2:
        base = np.abs(rng.standard_normal((max(2, n2 // int(rng.integers(2, 30))), 128)))
        d2 = base[rng.integers(0, len(base), n2)]; d1 = base[rng.integers(0, len(base), n1)]
    elif fam == 3:
        d1 = np.abs(rng.standard_normal((n1, 128))) * (rng.random((n1, 128)) < 0.15); d2 = np.abs(rng.standard_normal((n2, 128))) * (rng.random((n2, 128)) < 0.15)
    else:
        d1 = rng.standard_normal((n1, 128)) * 10.0 ** rng.uniform(-3, 1); d2 = rng.standard_normal((n2, 128)) * 10.0 ** rng.uniform(-3, 1)
    d1 = np.ascontiguousarray(d1, np.float32); d2 = np.ascontiguousarray(d2, np.float32)
    if fam != 4:
        d1 /= np.maximum(np.linalg.norm(d1, axis=1, keepdims=True), 1e-20); d2 /= np.maximum(np.linalg.norm(d2, axis=1, keepdims=True), 1e-20)
    f = run(d1, d2, S.MATCH_FUSED); e = run(d1, d2, S.MATCH_EXACT)
    bad = np.flatnonzero((f[0] != e[0]) | (f[1] != e[1]) | (f[2] != e[2]))
    if bad.size:
        print(f"MISMATCH trial {trials}: {n1} x {n2} family {fam}: {bad.size} queries, first {bad[:5]}", flush=True)
        np.savez("gpurun_out/match_fused_mismatch.npz", d1=d1, d2=d2)
        sys.exit(1)
    pairs += n1 * n2; trials += 1
print(f"match_fused soak: {trials} random matches, {pairs:.3e} (query, row) pairs, every best / second / index equal to the exact matcher's")
