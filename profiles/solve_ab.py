import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
import cuda_sfm_amd_ab as S            # the lab-bench flavour (make ab): switches, probes, traces
from cuda_sfm_amd_ab import synth
dev = torch.device("cuda:0"); ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
n, H = 4096, 1 << 20
sc = synth.two_view_scene(n)
d = torch.from_numpy(sc["sift"].view(np.uint8).reshape(n, 576)).to(dev)
pair = S.ImagePair(ctx, sc["K"], sc["Kinv"], 2, n); pair.fillXU(d)
for sweeps in (0, 7):
    for scalar in (0, 1):
        p = S.default_params(n, num_hypotheses=H, kernel=S.KERNEL_SPLIT, jacobi_sweeps=sweeps)
        p.reserved[0] = scalar
        for _ in range(3): pair.estimateE(p)
        ctx.synchronize(); ctx.kernel_timing(True)
        for _ in range(10): pair.estimateE(p)
        ctx.synchronize()
        solve_ms, score_ms, calls = ctx.kernel_timing_read(); ctx.kernel_timing(False)
        print("sweeps", sweeps, "scalar" if scalar else "packed", "solve ms", solve_ms / calls, "score ms", score_ms / calls, pair.get_best())
