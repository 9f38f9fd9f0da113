"""Replay a case dumped by tests/fuzz_gpu.py (FUZZ_DUMP=<dir>): estimateE on the product library and on the lab-bench library under the
pre-filter's forms, every count against the oracle.   python profiles/fuzz_case.py <case.npz>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cuda_sfm_amd as P
import cuda_sfm_amd_ab as A
import oracle as O
from cuda_sfm_amd import synth

d = np.load(sys.argv[1])
n, H, thr, seed = int(d["n"]), int(d["H"]), float(d["thr"]), int(d["seed"])
sift = d["sift"].view(synth.SIFT_DTYPE).reshape(-1)
dev = torch.device("cuda:0")
_, _, X0, X1 = O.fill_xu(sift, d["Kinv"])
for sweeps in (0, int(d["sweeps"])):
    key, ocounts, _ = O.ransac_range(X0, X1, 0, H, np.float32(thr), sweeps, seed=seed)
    for lib, name, r3 in ((P, "product", 0), (A, "lab bench, reserved[3] = 6 (per hypothesis always)", 6), (A, "lab bench, reserved[3] = 7 (per tile always)", 7), (A, "lab bench, reserved[3] = 5 (alignbit scan)", 5),
                          (A, "lab bench, reserved[3] = 3 (records from the stand-alone kernel whatever the solver)", 3)):
        ctx = lib.Context(0, torch.cuda.current_stream().cuda_stream)
        pair = lib.ImagePair(ctx, d["K"], d["Kinv"], 2, n)
        d_sift = torch.from_numpy(sift.view(np.uint8).reshape(n, 576).copy()).to(dev)
        pair.fillXU(d_sift)
        p = lib.default_params(n, num_hypotheses=H, seed=seed, kernel=lib.KERNEL_PREFILTER, jacobi_sweeps=sweeps, threshold=thr)
        if r3:
            p.reserved[3] = r3
        out = []
        for call in range(2):
            pair.estimateE(p)
            c = pair.get_inlier_counts(H)
            bad = np.flatnonzero(c != ocounts)
            out.append(f"call {call + 1}: rule {pair.last_launch().get('prefilter_rule')} kernel {pair.last_launch()['kernel']} bad {bad.size} (gpu - oracle in [{int((c - ocounts).min())}, {int((c - ocounts).max())}])")
        print(f"sweeps {sweeps} {name}: " + "; ".join(out), flush=True)
        pair.close()
