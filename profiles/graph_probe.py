import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    ctx = S.Context(0, st.cuda_stream)
    n = 4096
    scene = synth.two_view_scene(n)
    d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.fillXU(d_sift)
    for H in (131072, 1 << 20):
        p = S.default_params(n, num_hypotheses=H)
        for _ in range(20): pair.estimateE(p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): pair.estimateE(p)
        torch.cuda.synchronize(); plain = (time.perf_counter() - t0) / 100
        ref = pair.get_best()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=st):
                pair.estimateE(p)
            for _ in range(20): g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100): g.replay()
            torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 100
            print(H, "plain us", plain * 1e6, "graph us", gr * 1e6, pair.get_best() == ref)
        except Exception as e:
            print(H, "plain us", plain * 1e6, "graph capture failed:", repr(e)[:300])
