"""Time sfm_match with each matcher (exact / pre-filter / fused) at a list of sizes: ms per call, fifty calls back to back."""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
names = {S.MATCH_EXACT: "exact", S.MATCH_PREFILTER: "prefilter", S.MATCH_FUSED: "fused"}
for n in [int(x) for x in os.environ.get('SIZES', '512,1024,2048,3000,4096,5500,8192,16384').split(',')]:
    d1, d2, perm = synth.descriptors(n)
    s1 = synth.sift_records(d1); s2 = synth.sift_records(d2)
    t1 = torch.from_numpy(s1.view(np.uint8).reshape(n, 576)).to(dev); t2 = torch.from_numpy(s2.view(np.uint8).reshape(n, 576)).to(dev)
    row = []
    for k in (S.MATCH_EXACT, S.MATCH_PREFILTER, S.MATCH_FUSED):
        ctx.set_match_kernel(k)
        for _ in range(5): ctx.match(t1, n, t2, n)
        ctx.synchronize(); ctx.timer_start()
        for _ in range(50): ctx.match(t1, n, t2, n)
        ms = ctx.timer_stop() / 50
        row.append(f"{names[k]} {ms * 1e3:8.1f} us")
    print(f"{n:6d} x {n:<6d}", "   ".join(row), flush=True)
