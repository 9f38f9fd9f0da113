import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import cuda_sfm_amd as S
from helpers import read_pnm_grey, dino_frame, DINO_K, DINO_KINV, DINO_SIFT
dev = torch.device('cuda', 0)
views8 = [read_pnm_grey(dino_frame(k)).astype(np.uint8) for k in range(36)]
pairs = [(i, j) for i in range(36) for j in range(i + 1, 36)]
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
ts = []
for k in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res, counts = S.process_views(ctx, views8, DINO_K, DINO_KINV, pairs=pairs, max_pts=8192, sift=DINO_SIFT, device=dev)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(' '.join('%.1f' % t for t in ts))
