"""How much of a small estimateE (131072 hypotheses x 4096 matches = one of 8 ranks) can overlap with the next one?
Two Image_pairs on two contexts / streams, calls issued alternately with no dependency between them, against one stream."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
dev = torch.device("cuda:0")
n = 4096
scene = synth.two_view_scene(n)
d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
ctxs = [S.Context(0) for _ in range(3)]
for c in ctxs: c.own_stream()
pairs = [S.ImagePair(c, scene["K"], scene["Kinv"], 2, n) for c in ctxs]
for p in pairs: p.fillXU(d_sift)
for H in (131072, 1 << 20):
    prm = S.default_params(n, num_hypotheses=H)
    for k in (1, 2, 3):
        for _ in range(10):
            for i in range(k): pairs[i].estimateE(prm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 120
        for r in range(reps): pairs[r % k].estimateE(prm)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(H, "streams", k, "us per estimateE", round(dt * 1e6, 1))
