import os, sys, time, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
import cuda_sfm_amd as S
from helpers import read_pnm_grey, dino_frame, DINO_K, DINO_KINV, DINO_SIFT
dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
views = [read_pnm_grey(dino_frame(k)) for k in range(36)]
pairs = [(i, j) for i in range(36) for j in range(i + 1, 36)]
S.process_views(ctx, views[:9], DINO_K, DINO_KINV, max_pts=8192, sift=DINO_SIFT, device=dev)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res, counts = S.process_views(ctx, views, DINO_K, DINO_KINV, pairs=pairs, max_pts=8192, sift=DINO_SIFT, device=dev)
    torch.cuda.synchronize(); print("ms", 1e3 * (time.perf_counter() - t0))
import ctypes as C
if hasattr(S._lib, "sfm_debug_mf"):                   # built with -DSFM_MF_TRACE (see match_fused.hip)
    buf = (C.c_ulonglong * 32)()
    S._lib.sfm_debug_mf(buf)
    nq = sum(counts[i] for (i, j) in pairs)
    print("exact chains in all calls so far:", buf[30], "flushes:", buf[31], "queries x matches of one run:", nq, "-> chains per query and match:", buf[30] / 3.0 / nq)
