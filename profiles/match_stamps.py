"""Where the time of one exact-matcher launch goes: wall-clock stamps (100 MHz) written by thread 0 of every block of a lib
built with -DSFM_MATCH_STAMPS (a debug build, see profiles/r04_match_stamps.txt for the recipe)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
lib = ctypes.CDLL(os.path.join(os.environ["SFM_AMD_LIB_DIR"], "libsfm_amd.so"))
for n in (2048, 2155):
    d1, d2, perm = synth.descriptors(n)
    t1, t2 = torch.from_numpy(d1).to(dev), torch.from_numpy(d2).to(dev)
    best = torch.empty(n, dtype=torch.float32, device=dev); sec = torch.empty_like(best); idx = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.set_match_kernel(S.MATCH_EXACT)
    for _ in range(30): ctx.match_soa(t2, n, 128, t1, n, 128, best, sec, idx)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 1024, np.uint64)
    lib.sfm_dbg_match_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    st = buf.reshape(1024, 8).astype(np.int64)
    live = st[:, 0] > 0
    st = st[live]
    t0 = st[:, 0].min()
    rel = (st - t0) * 0.01                                 # us
    names = ["entry", "queries resident", "first stage staged", "main loop done", "partial stored", "ticket back", "merged (last blocks)", "emitted (last blocks)"]
    print(f"n = {n}: {len(st)} blocks")
    for k, nm in enumerate(names):
        col = rel[:, k][st[:, k] >= t0] if k < 6 else rel[:, k][st[:, k] > st[:, 5]]
        if len(col): print(f"  {nm:24s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us   ({len(col)} blocks)")
