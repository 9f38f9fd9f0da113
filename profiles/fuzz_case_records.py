"""The per-hypothesis records of a dumped fuzz case as the two record builders of the lab-bench library write them -- inline in the lane-solve kernel
(reserved[3] = 6) and by the stand-alone kernel (reserved[3] = 3 on the first call) -- plus the pair's bound words.   python profiles/fuzz_case_records.py <case.npz>"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cuda_sfm_amd_ab as A
import oracle as O
from cuda_sfm_amd import synth
hip = C.CDLL("libamdhip64.so")

def fetch(pair, which, dtype):
    p, b = pair.device_ptr(which)
    out = np.empty(b // np.dtype(dtype).itemsize, dtype)
    torch.cuda.synchronize()
    assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(p), C.c_size_t(b), 2) == 0
    return out

d = np.load(sys.argv[1])
n, H, thr, seed = int(d["n"]), int(d["H"]), float(d["thr"]), int(d["seed"])
sift = d["sift"].view(synth.SIFT_DTYPE).reshape(-1)
dev = torch.device("cuda:0")
_, _, X0, X1 = O.fill_xu(sift, d["Kinv"])
key, ocounts, _ = O.ransac_range(X0, X1, 0, H, np.float32(thr), 0, seed=seed)
recs = {}
for r3 in (6, 3):
    ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
    pair = A.ImagePair(ctx, d["K"], d["Kinv"], 2, n)
    pair.fillXU(torch.from_numpy(sift.view(np.uint8).reshape(n, 576).copy()).to(dev))
    p = A.default_params(n, num_hypotheses=H, seed=seed, kernel=A.KERNEL_PREFILTER, jacobi_sweeps=0, threshold=thr)
    p.reserved[3] = r3
    pair.estimateE(p)
    c = pair.get_inlier_counts(H)
    recs[r3] = fetch(pair, 100, np.uint16).reshape(H, 32)
    bound = fetch(pair, 101, np.uint64)
    print(f"reserved[3] = {r3}: rule {pair.last_launch()['prefilter_rule']} bad {(c != ocounts).sum()}; bound words:", [hex(int(w)) for w in bound])
    bad = np.flatnonzero(c != ocounts)
diff = np.flatnonzero((recs[6] != recs[3]).any(axis=1))
print("records that differ between the two builders:", diff.size, "of", H, "; hypotheses with wrong counts under the stand-alone builder:", bad.size, "; of those with differing records:", np.intersect1d(diff, bad).size)
for h in diff[:6]:
    a, b = recs[6][h].view(np.float16).astype(np.float32), recs[3][h].view(np.float16).astype(np.float32)
    print("hyp", h, "inline    ", np.array2string(a[:11], precision=4), "flags", hex(int(recs[6][h][25])))
    print("hyp", h, "standalone", np.array2string(b[:11], precision=4), "flags", hex(int(recs[3][h][25])), " ratio of the first slots", b[0] / a[0] if a[0] else None)
