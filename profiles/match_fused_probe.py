"""One size, the fused matcher only (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
n = int(os.environ.get("N", "2048"))
d1, d2, perm = synth.descriptors(n)
s1 = synth.sift_records(d1); s2 = synth.sift_records(d2)
t1 = torch.from_numpy(s1.view(np.uint8).reshape(n, 576)).to(dev); t2 = torch.from_numpy(s2.view(np.uint8).reshape(n, 576)).to(dev)
ctx.set_match_kernel(int(os.environ.get("KERNEL", str(S.MATCH_FUSED))))
for _ in range(20): ctx.match(t1, n, t2, n)
ctx.synchronize()
import ctypes as C
if hasattr(S._lib, "sfm_debug_mf"):                   # built with -DSFM_MF_TRACE (see match_fused.hip)
    buf = (C.c_ulonglong * 32)()
    S._lib.sfm_debug_mf(buf)
    v = list(buf)
    names = {0: "start", 1: "queries stored", 2: "fragments read", 21: "stages done", 22: "last chains done", 23: "ticket"}
    for st in range(3):
        for k, nm in enumerate(("begin", "stored", "barrier", "mfma issued", "masks", "appended")):
            names[3 + 6 * st + k] = f"s{st} {nm}"
    print("cycles from start:", {names[i]: int(v[i] - v[0]) for i in sorted(names) if v[i]})
