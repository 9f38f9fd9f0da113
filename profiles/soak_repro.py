"""Re-runs the configurations a prefilter_soak.py run flagged (it records n, flavour, seed, thr, sweeps but not the noise /
outlier / focal draws: every combination is tried) and dumps what differs: hypothesis, both counts, its E, the points.
    python profiles/soak_repro.py '<json mismatches list>' out.npz"""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd_ab as S            # the lab-bench flavour (make ab): switches, probes, traces
from cuda_sfm_amd_ab import synth

cases = json.loads(sys.argv[1])
dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
H = 1 << 18
found = {}
for ci, c in enumerate(cases):
    n, flavour, seed, thr, sweeps = c["n"], c["flavour"], c["seed"], np.float32(c["thr"]), c["sweeps"]
    focals = [200.0, 500.0] if flavour == "wide" else [9000.0] if flavour == "narrow" else [2360.0]
    outl = [0.0] if flavour == "clean" else [0.2, 0.5, 0.8]
    for focal, noise, of in itertools.product(focals, [0.0, 0.3, 1.5], outl):
        sc = synth.two_view_scene(n, seed=seed, focal=focal, noise_px=noise, outlier_frac=of)
        d_sift = torch.from_numpy(sc["sift"].view(np.uint8).reshape(n, 576)).to(dev)
        pair = S.ImagePair(ctx, sc["K"], sc["Kinv"], 2, n)
        pair.fillXU(d_sift)
        res = []
        for kernel in (S.KERNEL_SPLIT, S.KERNEL_PREFILTER):
            p = S.default_params(n, num_hypotheses=H, seed=seed & 0xFFFF, kernel=kernel, threshold=float(thr), jacobi_sweeps=sweeps)
            pair.estimateE(p)
            res.append(pair.get_inlier_counts(H).copy())
        diff = np.nonzero(res[0] != res[1])[0]
        if len(diff):
            Ec = pair.get_E_candidates(H)
            X0 = pair.get_XU(S.BUF_X0) if hasattr(S, "BUF_X0") else pair.get_XU(0)
            X1 = pair.get_XU(S.BUF_X1) if hasattr(S, "BUF_X1") else pair.get_XU(1)
            for h in diff[:4]:
                print(json.dumps({"case": ci, "n": n, "focal": focal, "noise": noise, "outliers": of, "thr": float(thr), "sweeps": sweeps,
                                  "hyp": int(h), "count_split": int(res[0][h]), "count_prefilter": int(res[1][h]), "E": [float(x) for x in Ec[h].ravel()]}), flush=True)
            found["c%d_X0" % ci] = X0; found["c%d_X1" % ci] = X1; found["c%d_E" % ci] = Ec[diff[:4]]
            found["c%d_thr" % ci] = np.float32(thr); found["c%d_hyp" % ci] = diff[:4]
        pair.close()
np.savez_compressed(sys.argv[2], **found)
print("saved", sorted(found))
