"""Where block 0 of ransac_score_prefilter spends its time (sfm_ransac_last_phases), per shard size and grid width."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cuda_sfm_amd_ab as S            # the lab-bench flavour (make ab): switches, probes, traces
from cuda_sfm_amd_ab import synth

dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
n = 4096
scene = synth.two_view_scene(n)
d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
pair.fillXU(d_sift)
for H, cols in ((131072, 0), (131072, 64), (262144, 0), (1048576, 0), (1048576, 64)):
    p = S.default_params(n, num_hypotheses=H, seed=3, kernel=S.KERNEL_PREFILTER)
    p.reserved[2] = cols
    for _ in range(10):
        pair.estimateE(p)
    ctx.kernel_timing(True) if hasattr(ctx, "kernel_timing") else None
    pair.estimateE(p)
    t = pair.last_phases()
    us = lambda k: round(t[k] / 100.0, 2)
    print(json.dumps({"hypotheses": H, "grid": pair.last_launch()["grid"], "block0_lifetime_us": us(1), "tile_staged_us": us(2), "first_pass_prepared_us": us(3),
                      "first_block_scanned_us": us(4), "ring_drained_us": us(7), "first_pass_done_us": us(5), "passes_of_wave0": t[6],
                      "shader_mhz": round(100.0 * t[0] / max(t[1], 1))}), flush=True)
