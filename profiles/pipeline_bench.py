#!/usr/bin/env python3
"""Stage timings of the two-view pipeline at the reference's published operating point
(dino pair, ~2k features; img/data.xlsx: match 1.484, fillXU 0.0499, estimateE 24.12,
candidates 0.5996, choosePose 6.0118, triangulation 8.1359 ms on a GTX 1080 Ti).
Synthetic dino-shaped inputs (SIFT extraction is out of scope).  Run on the GPU box."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth

REF_MS = {"match": 1.484, "fillXU": 0.0499, "estimateE": 24.12, "posecandidates": 0.5996, "choosePose": 6.0118, "triangulation": 8.1359}
dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


for n, H in ((2048, 256), (2048, 1024), (16384, 65536)):
    sc = synth.two_view_scene(n)
    d1, d2, perm = synth.descriptors(n)
    s1 = synth.sift_records(d1); s1["xpos"], s1["ypos"] = sc["sift"]["xpos"], sc["sift"]["ypos"]
    s2 = synth.sift_records(d2); s2["xpos"], s2["ypos"] = sc["sift"]["match_xpos"][perm], sc["sift"]["match_ypos"][perm]
    t1 = torch.from_numpy(s1.view(np.uint8).reshape(n, 576)).to(dev)
    t2 = torch.from_numpy(s2.view(np.uint8).reshape(n, 576)).to(dev)
    pair = S.ImagePair(ctx, sc["K"], sc["Kinv"], 2, n)
    p = S.default_params(n, num_hypotheses=H)
    out = {"n": n, "hypotheses": H}
    out["match"] = timed(lambda: ctx.match(t1, n, t2, n))
    out["fillXU"] = timed(lambda: pair.fillXU(t1))
    out["estimateE"] = timed(lambda: pair.estimateE(p))
    out["posecandidates"] = timed(lambda: pair.computePosecandidates())
    out["choosePose"] = timed(lambda: pair.choosePose())
    out["triangulation"] = timed(lambda: pair.linear_triangulation())
    out["total"] = sum(out[k] for k in REF_MS)
    if n == 2048:
        out["speedup_vs_published_1080Ti"] = {k: REF_MS[k] / out[k] for k in REF_MS}
    hyp, cnt = pair.get_best()
    out["inliers"] = cnt
    print(json.dumps(out))
