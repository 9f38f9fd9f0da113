"""FindHomography (row f2) timing: whole call (gate on the host, ~10 launches, 40-byte read-back).
Run on the GPU box: python profiles/homography_bench.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth

dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
for n in (2048, 8192):
    sc = synth.homography_scene(n, seed=7)
    d = torch.from_numpy(sc["sift"].view(np.uint8).reshape(n, 576)).to(dev)
    for loops in (1000, 10000):
        for _ in range(5):
            H, nm = ctx.find_homography(d, n, num_loops=loops, min_score=0.0, max_ambiguity=1.0, thresh=5.0, seed=3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            H, nm = ctx.find_homography(d, n, num_loops=loops, min_score=0.0, max_ambiguity=1.0, thresh=5.0, seed=3)
        ms = (time.perf_counter() - t0) / 30 * 1e3
        print(json.dumps({"points": n, "loops": loops, "ms": round(ms, 4), "support": int(nm)}))
