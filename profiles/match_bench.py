#!/usr/bin/env python3
"""Times the MFMA descriptor matcher (sfm_match_soa) on synthetic CudaSift-like descriptors and
prints achieved TFLOP/s against the 157.3 TFLOP/s f32 MFMA peak, next to the CPU oracle matcher
(restating MatchC1, CudaSift/match.cu:57-71, OpenMP over queries) on the box's host cores.  Run on the GPU box."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth

dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
for n in (2048, 16384):
    d1, d2, perm = synth.descriptors(n)
    t1, t2 = torch.from_numpy(d1).to(dev), torch.from_numpy(d2).to(dev)
    best = torch.empty(n, dtype=torch.float32, device=dev); sec = torch.empty_like(best)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    for _ in range(3):
        ctx.match_soa(t2, n, 128, t1, n, 128, best, sec, idx)
    torch.cuda.synchronize()
    reps = 20
    ctx.timer_start()
    for _ in range(reps):
        ctx.match_soa(t2, n, 128, t1, n, 128, best, sec, idx)
    ms = ctx.timer_stop() / reps
    flops = 2.0 * n * n * 128
    ok = float((idx.cpu().numpy() == perm).mean())
    row = {"n": n, "ms": ms, "tflops": flops / ms / 1e9, "frac_of_157.3": flops / ms / 1e9 / 157.3, "perm_recovered": ok}
    if not os.environ.get("MATCH_NO_CPU"):
        import time
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        cores = len(os.sched_getaffinity(0))
        t0 = time.perf_counter()
        cb, cs, ci = O.match_desc(d2, d1, nthreads=cores)
        row["cpu_oracle_ms"] = 1e3 * (time.perf_counter() - t0); row["cpu_cores"] = cores
        row["cpu_index_equal"] = bool(np.array_equal(ci, idx.cpu().numpy()))
    print(json.dumps(row))
