"""BASELINE configs[4] on ONE GPU: 36 views (720x576, the dino frame size) -> ExtractSift per view -> ring pairs
(36) and all unordered pairs (630): MatchSiftData + fillXU + estimateE (H = N/8) + pose + triangulation per pair
(cuda_sfm_amd.process_views; with --gpus N the views and pairs are dealt round-robin over the ranks).
Synthetic views (one textured scene, camera sliding along x).  Run on the GPU box: python profiles/ring_bench.py"""
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

dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
V, w, h = 36, 720, 576
base = np.array([5.0, 8.0, 12.0, 16.0, 7.0, 10.0], np.float32)
views = [synth.stereo_pair(w, h, seed=9, disparities=tuple((0.1 * k) * base))[1] if k else synth.stereo_pair(w, h, seed=9)[0] for k in range(V)]
K, Kinv = synth.camera(w, h)
sift = dict(num_octaves=5, init_blur=1.5, thresh=1.0)          # src/main.cpp:267-277
runs_todo = [("ring_36_pairs", views, S.ring_pairs(V)), ("all_630_pairs", views, [(i, j) for i in range(V) for j in range(i + 1, V)])]
# the reference's own sequence (data/dino/viff.000-035.ppm as 8-bit grey fixtures): BASELINE configs[4] literally
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import read_pnm_grey, dino_frame
if os.path.exists(dino_frame(35)):
    dino = [read_pnm_grey(dino_frame(k)) for k in range(36)]
    runs_todo += [("dino_ring_36_pairs", dino, S.ring_pairs(36)), ("dino_all_630_pairs", dino, [(i, j) for i in range(36) for j in range(i + 1, 36)])]
    dino8 = [d.astype(np.uint8) for d in dino]           # the frames as the 8-bit images they are (sfm_extract_views_u8)
    runs_todo += [("dino_ring_36_pairs_8bit_images", dino8, S.ring_pairs(36)), ("dino_all_630_pairs_8bit_images", dino8, [(i, j) for i in range(36) for j in range(i + 1, 36)])]
for name, views, pairs in runs_todo:
    S.process_views(ctx, views[:9], K, Kinv, max_pts=8192, sift=sift, device=dev)      # warm-up (buffers, lanes, clocks)
    runs = []
    for _ in range(3):                                   # the first run still grows buffers (pooled pairs, records, allocator)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res, counts = S.process_views(ctx, views, K, Kinv, pairs=pairs, max_pts=8192, sift=sift, device=dev)
        torch.cuda.synchronize(); runs.append(time.perf_counter() - t0)
    dt = min(runs)
    inl = [int(r[26]) for r in res.values()]
    print(json.dumps({name: {"views": V, "pairs": len(pairs), "done": len(res), "features_per_view": [min(counts), max(counts)],
                             "ms_total": 1e3 * dt, "ms_runs": [round(1e3 * r, 3) for r in runs], "ms_per_pair": 1e3 * dt / len(pairs), "median_inliers": int(np.median(inl)),
                             "note": "host images -> device inside the timed region (PCIe-inclusive); per-pair work inside sfm_process_pairs (C)"}}))
