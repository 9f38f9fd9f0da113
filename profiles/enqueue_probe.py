"""How long does the HOST take to enqueue one pipelined estimateE step, against how long the device takes to run it?
(If the two are close at small shards the step is launch-bound, not kernel-bound.)  One JSON line per shard size."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth

dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
n = 4096
scene = synth.two_view_scene(n)
d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
pair.fillXU(d_sift)
for H in (16384, 131072, 262144, 1048576):
    p = S.default_params(n, num_hypotheses=H, seed=3)
    for _ in range(30):
        pair.estimateE_pipelined(p)
    pair.flush(); torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        pair.estimateE_pipelined(p)
    t1 = time.perf_counter()
    pair.flush(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    # serial calls for comparison
    t3 = time.perf_counter()
    for _ in range(K):
        pair.estimateE(p)
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    print(json.dumps({"hypotheses": H, "pipelined_enqueue_us_per_step": round((t1 - t0) / K * 1e6, 1), "pipelined_total_us_per_step": round((t2 - t0) / K * 1e6, 1),
                      "serial_enqueue_us_per_step": round((t4 - t3) / K * 1e6, 1), "serial_total_us_per_step": round((t5 - t3) / K * 1e6, 1)}), flush=True)
