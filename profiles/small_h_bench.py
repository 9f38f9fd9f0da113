#!/usr/bin/env python3
"""estimateE latency vs hypothesis count for the two kernel families (which one should AUTO pick?)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
def timed(fn, reps=30):
    for _ in range(3): fn()
    ctx.synchronize(); ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop() / reps
for n in (2048, 4096):
    sc = synth.two_view_scene(n)
    t1 = torch.from_numpy(sc["sift"].view(np.uint8).reshape(n, 576)).to(dev)
    pair = S.ImagePair(ctx, sc["K"], sc["Kinv"], 2, n); pair.fillXU(t1)
    for H in (256, 1024, 4096, 16384, 65536, 262144):
        row = {"n": n, "H": H}
        for name, k in (("split", S.KERNEL_SPLIT), ("fused", S.KERNEL_FUSED)):
            p = S.default_params(n, num_hypotheses=H, kernel=k)
            row[name + "_ms"] = round(timed(lambda: pair.estimateE(p)), 4)
        print(json.dumps(row))
