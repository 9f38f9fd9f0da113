"""sfm_estimate_E_pipelined (two slots on ONE Image_pair) against plain sfm_estimate_E: context on the NULL stream vs a
stream of its own, kernel timing events on / off."""
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
torch.cuda.synchronize()
for own in (False, True):
    ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
    if own: ctx.own_stream()
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.fillXU(d_sift); ctx.synchronize()
    for H in (131072, 1 << 20):
        prm = S.default_params(n, num_hypotheses=H)
        for timing in (False, True):
            for mode in ("serial", "pipelined"):
                f = pair.estimateE if mode == "serial" else pair.estimateE_pipelined
                for _ in range(20): f(prm)
                pair.flush(); ctx.synchronize(); torch.cuda.synchronize()
                ctx.kernel_timing(timing)
                t0 = time.perf_counter()
                for _ in range(100): f(prm)
                pair.flush(); ctx.synchronize(); torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 100
                ctx.kernel_timing(False)
                print("own_stream" if own else "null_stream", H, "timing" if timing else "no-timing", mode, round(dt * 1e6, 1), "us")
