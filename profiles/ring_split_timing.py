import os, sys, time, json
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
V, w, h = 36, 720, 576
base = np.array([5.0, 8.0, 12.0, 16.0, 7.0, 10.0], np.float32)
views = [synth.stereo_pair(w, h, seed=9, disparities=tuple((0.1 * k) * base))[1] if k else synth.stereo_pair(w, h, seed=9)[0] for k in range(V)]
K, Kinv = synth.camera(w, h)
sift = dict(num_octaves=5, init_blur=1.5, thresh=1.0)
orig = S.process_pairs_local
def timed_ppl(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(*a, **k)
    torch.cuda.synchronize(); print("   process_pairs_local ms", 1e3 * (time.perf_counter() - t0), "pairs", len(a[1]))
    return r
S.process_pairs_local = timed_ppl
for pairs in (S.ring_pairs(V), [(i, j) for i in range(V) for j in range(i + 1, V)]):
    S.process_views(ctx, views[:9], K, Kinv, max_pts=8192, sift=sift, device=dev)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res, counts = S.process_views(ctx, views, K, Kinv, pairs=pairs, max_pts=8192, sift=sift, device=dev)
        torch.cuda.synchronize(); print("total ms", 1e3 * (time.perf_counter() - t0))
