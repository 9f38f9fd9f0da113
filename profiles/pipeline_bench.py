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
    # the three pose calls as ONE launch (sfm_pose_chain; what sfm_process_pairs runs per pair)
    out["pose_chain"] = timed(lambda: pair.pose_chain())
    out["total_with_pose_chain"] = out["match"] + out["fillXU"] + out["estimateE"] + out["pose_chain"]
    if n == 2048:
        out["speedup_vs_published_1080Ti"] = {k: REF_MS[k] / out[k] for k in REF_MS}
    hyp, cnt = pair.get_best()
    out["inliers"] = cnt
    print(json.dumps(out))


# ---- the whole application run of the reference (src/main.cpp:249-307; img/data.xlsx total 47.36 ms incl. 6.96 ms SIFT,
# dino frames 720x576): two images -> ExtractSift x2 -> match -> fillXU -> estimateE -> poses -> triangulation, synthetic
# stereo pair of that size, everything resident on the device, default parameters of main.cpp (thresh 1.0, initBlur 1.5).
def whole_run(a, b, label, extra):
    h, w = a.shape
    p = (w + 127) // 128 * 128

    def dev_img(img):
        pad = np.zeros((h, p), np.float32); pad[:, :w] = img
        return torch.from_numpy(pad).to(dev)

    da, db = dev_img(a), dev_img(b)
    s1 = torch.zeros((32768, 576), dtype=torch.uint8, device=dev); s2 = torch.zeros_like(s1)
    L = S.sift_temp_layout(w, h, 5, False)
    tmp = torch.zeros(L.total_floats, dtype=torch.float32, device=dev)
    K, Kinv = synth.camera(w, h)                        # main.cpp:292-297: f = 2360, principal point = image centre
    state = {}

    def whole():
        n1, _ = ctx.extract_sift(s1, 32768, da, w, h, p, 5, 1.5, 1.0, 0.0, False, tmp)
        n2, _ = ctx.extract_sift(s2, 32768, db, w, h, p, 5, 1.5, 1.0, 0.0, False, tmp)
        ctx.match(s1, n1, s2, n2)
        pr = state.get(n1)
        if pr is None:
            pr = state[n1] = S.ImagePair(ctx, K, Kinv, 2, n1)
        pr.fillXU(s1)
        pr.estimateE(S.default_params(n1))
        pr.pose_chain()
        return n1, n2, pr

    n1, n2, pr = whole()
    ms = timed(whole, reps=30)

    # the same run with the two extractions issued together (sfm_extract_sift_begin / _end on two contexts)
    ctx2 = state.get("ctx2")
    if ctx2 is None:
        ctx2 = state["ctx2"] = S.Context(0); ctx2.own_stream()
    tmp2 = torch.zeros(L.total_floats, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def whole_overlapped():
        ctx.extract_sift_begin(s1, 32768, da, w, h, p, 5, 1.5, 1.0, 0.0, False, tmp)
        ctx2.extract_sift_begin(s2, 32768, db, w, h, p, 5, 1.5, 1.0, 0.0, False, tmp2)
        m1, _ = ctx.extract_sift_end(); m2, _ = ctx2.extract_sift_end()
        ctx.match(s1, m1, s2, m2)
        pr = state[m1]
        pr.fillXU(s1)
        pr.estimateE(S.default_params(m1))
        pr.pose_chain()
        return m1, m2

    assert whole_overlapped() == (n1, n2)
    ms_ovl = timed(whole_overlapped, reps=30)
    sift_ms = timed(lambda: (ctx.extract_sift(s1, 32768, da, w, h, p, 5, 1.5, 1.0, 0.0, False, tmp),
                             ctx.extract_sift(s2, 32768, db, w, h, p, 5, 1.5, 1.0, 0.0, False, tmp)), reps=30)
    _, cnt = pr.get_best()
    out = {"features": [n1, n2], "hypotheses": n1 // 8, "inliers": cnt, "ms": ms, "ms_pair_extraction_overlapped": ms_ovl, "sift_x2_ms": sift_ms,
           "published_ms_1080Ti": 47.36, "published_sift_ms": 6.96, "speedup": 47.36 / ms, "speedup_overlapped": 47.36 / ms_ovl}
    out.update(extra)
    print(json.dumps({label: out}))


a, b, _, _ = synth.stereo_pair(720, 576, seed=5)
whole_run(a, b, "whole_run_720x576", {"input": "synthetic stereo pair"})
# the reference program's own input (src/main.cpp:250-251), kept as a fixture: the published 47.36 ms are for THIS pair
dino = os.path.join(ROOT, "tests", "golden", "dino")
if os.path.exists(os.path.join(dino, "dino_grey_000.pgm")):
    sys.path.insert(0, os.path.join(ROOT, "tests")); from helpers import read_pnm_grey
    whole_run(read_pnm_grey(os.path.join(dino, "dino_grey_000.pgm")), read_pnm_grey(os.path.join(dino, "dino_grey_001.pgm")),
              "whole_run_dino_pair", {"input": "data/dino/viff.000.ppm + viff.001.ppm (the reference program's input, as 8-bit grey)"})
