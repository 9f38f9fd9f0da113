"""The front end of configs[4] alone: sfm_extract_views(_u8) on the 36 dino frames, ms per call (five calls)."""
import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import torch
import cuda_sfm_amd as S
from helpers import read_pnm_grey, dino_frame, DINO_SIFT
dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
NV = int(os.environ.get("NV", "36"))
views = [read_pnm_grey(dino_frame(k)) for k in range(NV)]
h, w = views[0].shape
max_pts = 8192; rec_bytes = max_pts * 576
for u8 in (False, True):
    imgs = [np.ascontiguousarray(v, np.uint8 if u8 else np.float32) for v in views]
    ct = C.c_ubyte if u8 else C.c_float
    ptrs = (C.POINTER(ct) * NV)(*[im.ctypes.data_as(C.POINTER(ct)) for im in imgs])
    block = torch.zeros((NV, rec_bytes + 64), dtype=torch.uint8, device=dev)
    cnts = (C.c_int * NV)()
    fn = S._lib.sfm_extract_views_u8 if u8 else S._lib.sfm_extract_views
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = fn(ctx._h, ptrs, NV, w, h, 0, 1, block.data_ptr(), rec_bytes + 64, max_pts, int(DINO_SIFT["num_octaves"]), float(DINO_SIFT["init_blur"]),
                float(DINO_SIFT["thresh"]), float(DINO_SIFT.get("lowest_scale", 0.0)), 0, cnts)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        assert rc == 0
    print("8-bit" if u8 else "float", "images: sfm_extract_views ms per call", [round(t, 3) for t in ts], "features", min(cnts), max(cnts))
