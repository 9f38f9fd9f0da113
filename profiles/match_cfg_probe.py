import os, sys, time, json
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
for n in [int(x) for x in os.environ.get('SIZES', '3000,4500,5500,7000,9000').split(',')]:
    d1, d2, perm = synth.descriptors(n)
    s1 = synth.sift_records(d1); s2 = synth.sift_records(d2)
    t1 = torch.from_numpy(s1.view(np.uint8).reshape(n, 576)).to(dev); t2 = torch.from_numpy(s2.view(np.uint8).reshape(n, 576)).to(dev)
    for _ in range(5): ctx.match(t1, n, t2, n)
    ctx.synchronize(); ctx.timer_start()
    for _ in range(50): ctx.match(t1, n, t2, n)
    ms = ctx.timer_stop() / 50
    print(os.environ.get("SFM_MATCH_CFG", "default"), n, round(ms, 4), "TF", round(2 * n * n * 128 / ms / 1e9, 1))
