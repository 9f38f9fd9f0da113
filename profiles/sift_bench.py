#!/usr/bin/env python3
"""ExtractSift timing on the GPU box.  Synthetic 8-bit images (cuda-sfm_amd/synth.py),
the reference's demo settings (5 octaves, initBlur 1.0, thresh 3.0; mainSift.cpp:58-66) at the sizes of
its README table (1280x960, 1920x1080; published 0.42 / 0.56 ms on an RTX 2080 Ti, 0.58 / 0.80 ms on a
GTX 1080 Ti, prefilter excluded) and the dino frame size (720x576; 6.96 ms published in img/data.xlsx).
Times are whole calls of sfm_extract_sift: low pass + pyramid + detection + orientation + descriptors +
the final counter read-back, images already on the device."""
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
REPS = int(os.environ.get("SIFT_REPS", "50"))
for (w, h, blobs) in ((720, 576, 500), (1280, 960, 1500), (1920, 1080, 2500)):
    img = synth.image(w, h, seed=11, blobs=blobs)
    p = (w + 127) // 128 * 128
    pad = np.zeros((h, p), np.float32); pad[:, :w] = img
    d_img = torch.from_numpy(pad).to(dev)
    d_sift = torch.zeros((32768, 576), dtype=torch.uint8, device=dev)
    L = S.sift_temp_layout(w, h, 5, False)
    d_temp = torch.zeros(L.total_floats, dtype=torch.float32, device=dev)
    for thresh in (3.0,):
        for _ in range(3):
            n, stored = ctx.extract_sift(d_sift, 32768, d_img, w, h, p, 5, 1.0, thresh, 0.0, False, d_temp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(REPS):
            n, stored = ctx.extract_sift(d_sift, 32768, d_img, w, h, p, 5, 1.0, thresh, 0.0, False, d_temp)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / REPS
        # algorithmic HBM bytes of the DoG stage: read the level once, write 7 planes (4 + 28 B / pixel)
        pix = sum(L.width[l] * L.height[l] for l in range(5))
        print(json.dumps({"image": f"{w}x{h}", "thresh": thresh, "num_pts": n, "stored": stored, "ms_per_image": round(ms, 4),
                          "mpix_per_s": round(w * h / ms / 1e3, 1), "pyramid_pixels": pix, "dog_bytes": 32 * pix}))
