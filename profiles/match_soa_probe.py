import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
for n in (2560, 4096, 5500, 8192, 16384):
    d1, d2, perm = synth.descriptors(n)
    t1, t2 = torch.from_numpy(d1).to(dev), torch.from_numpy(d2).to(dev)
    best = torch.empty(n, dtype=torch.float32, device=dev); sec = torch.empty_like(best); idx = torch.empty(n, dtype=torch.int32, device=dev)
    row = []
    for k, nm in ((S.MATCH_EXACT, "exact"), (S.MATCH_PREFILTER, "prefilter"), (S.MATCH_FUSED, "fused")):
        ctx.set_match_kernel(k)
        for _ in range(5): ctx.match_soa(t2, n, 128, t1, n, 128, best, sec, idx)
        ctx.synchronize(); ctx.timer_start()
        for _ in range(40): ctx.match_soa(t2, n, 128, t1, n, 128, best, sec, idx)
        row.append("%s %7.1f us" % (nm, ctx.timer_stop() / 40 * 1e3))
    print("%6d^2 plain arrays (ld 128):" % n, "  ".join(row), flush=True)
