import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
import oracle as O
from helpers import make_pair
dev = torch.device('cuda', 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
for n, H in ((16384, 131072), (16384, 131072), (4096, 131072), (16384, 65536), (8192, 131072)):
    scene = synth.two_view_scene(n, seed=77)
    pair, _ = make_pair(S, (torch, dev, ctx), scene)
    p = S.default_params(n, num_hypotheses=H, seed=11, kernel=S.KERNEL_PREFILTER)
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    key, oc, _ = O.ransac_range_fast(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed)
    for it in range(3):
        pair.ransac_score(p)
        c = pair.get_inlier_counts(H)
        bad = np.flatnonzero(c != oc)
        print(n, H, it, 'bad', bad.size, 'first', bad[:6], 'gpu', c[bad[:6]], 'oracle', oc[bad[:6]], 'passes hit', np.unique(bad // 32)[:10], 'sum diff', int((c.astype(np.int64) - oc).sum()))
    pair.close()
