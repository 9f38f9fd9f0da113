"""Full-size parity soak of the matrix-core pre-filter: random scenes (field of view, noise, outliers, forward motion with
correspondences on the epipole, duplicated points), 1100..14000 matches x 2^18 hypotheses each, EVERY count of
SFM_KERNEL_PREFILTER -- both forms: the first call after the fillXU scores with per-hypothesis operands, the second with per-tile operands
over the ordered copy -- against SFM_KERNEL_SPLIT (which the test-suite pins to the oracle), plus key / E / mask.
    python profiles/prefilter_soak.py [seconds] [seed]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
t_end = time.time() + budget
rounds, pairs_checked, bad = 0, 0, []
while time.time() < t_end:
    n = int(rng.choice([4096, 5000, 8192, 12000])) if rng.random() < 0.4 else int(rng.integers(1100, 14000))      # (round 6: any tile count, ragged last tiles)
    H = 1 << 18
    flavour = str(rng.choice(["plain", "wide", "narrow", "forward", "dup", "clean"]))
    focal = {"wide": float(rng.choice([200.0, 500.0])), "narrow": 9000.0}.get(flavour, 2360.0)
    seed = int(rng.integers(1, 1 << 30))
    sc = synth.two_view_scene(n, seed=seed, focal=focal, noise_px=float(rng.choice([0.0, 0.3, 1.5])),
                              outlier_frac=0.0 if flavour == "clean" else float(rng.choice([0.2, 0.5, 0.8])))
    s1 = sc["sift"]
    if flavour == "forward":
        c = np.array([360.0, 288.0]); d = np.stack([s1["xpos"], s1["ypos"]], 1) - c
        s1["match_xpos"], s1["match_ypos"] = (c + 1.05 * d).T.astype(np.float32)
        on = rng.integers(0, n, n // 50)
        for f, v in (("xpos", 360.0), ("ypos", 288.0), ("match_xpos", 360.0), ("match_ypos", 288.0)):
            s1[f][on] = v
    if flavour == "dup":
        src = rng.integers(0, n, n // 4); dst = rng.integers(0, n, n // 4)
        for f in ("xpos", "ypos", "match_xpos", "match_ypos"):
            s1[f][dst] = s1[f][src]
    thr = float(np.float32(10.0 ** rng.uniform(-8, -3)))
    d_sift = torch.from_numpy(s1.view(np.uint8).reshape(n, 576)).to(dev)
    pair = S.ImagePair(ctx, sc["K"], sc["Kinv"], 2, n)
    pair.fillXU(d_sift)
    res = []
    for kernel in (S.KERNEL_SPLIT, S.KERNEL_PREFILTER, S.KERNEL_PREFILTER):
        p = S.default_params(n, num_hypotheses=H, seed=seed & 0xFFFF, kernel=kernel, threshold=thr, jacobi_sweeps=int(rng.choice([0, 0, 7])) if kernel == S.KERNEL_SPLIT else res_sweeps)
        res_sweeps = p.jacobi_sweeps
        pair.estimateE(p)
        assert pair.last_launch()["kernel"] == kernel
        if kernel == S.KERNEL_PREFILTER:
            assert pair.last_launch()["prefilter_rule"] == (S.PREFILTER_PER_HYPOTHESIS if len(res) == 1 else S.PREFILTER_PER_TILE)
        res.append((pair.get_inlier_counts(H).copy(), pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy()))
    rounds += 1
    for form, r in (("per hypothesis", res[1]), ("per tile", res[2])):
        same = np.array_equal(res[0][0], r[0]) and res[0][1] == r[1] and np.array_equal(res[0][2].view(np.uint32), r[2].view(np.uint32)) and np.array_equal(res[0][3], r[3])
        pairs_checked += n * H
        if not same:
            bad.append({"n": n, "form": form, "flavour": flavour, "seed": seed, "thr": thr, "sweeps": res_sweeps, "differing_counts": int((res[0][0] != r[0]).sum())})
    pair.close()
print(json.dumps({"seconds": budget, "rounds": rounds, "pairs_checked": pairs_checked, "mismatches": bad}))
