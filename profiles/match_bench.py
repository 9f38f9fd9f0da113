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
SIZES = tuple(int(x) for x in os.environ.get("MATCH_SIZES", "2048,16384").split(","))
NAMES = {S.MATCH_EXACT: "exact fp32 MFMA", S.MATCH_PREFILTER: "fp16 MFMA pre-filter + exact candidates"}
for n in SIZES:
    d1, d2, perm = synth.descriptors(n)
    t1, t2 = torch.from_numpy(d1).to(dev), torch.from_numpy(d2).to(dev)
    best = torch.empty(n, dtype=torch.float32, device=dev); sec = torch.empty_like(best)
    idx = torch.empty(n, dtype=torch.int32, device=dev)

    def timed(kernel):
        ctx.set_match_kernel(kernel)
        for _ in range(3):
            ctx.match_soa(t2, n, 128, t1, n, 128, best, sec, idx)
        torch.cuda.synchronize()
        reps = 20
        ctx.timer_start()
        for _ in range(reps):
            ctx.match_soa(t2, n, 128, t1, n, 128, best, sec, idx)
        ms_ = ctx.timer_stop() / reps
        return ms_, ctx.last_match_kernel(), (best.cpu().numpy().copy(), sec.cpu().numpy().copy(), idx.cpu().numpy().copy())

    ms_exact, _, res_exact = timed(S.MATCH_EXACT)
    ms_pf, _, res_pf = timed(S.MATCH_PREFILTER)
    ms, ran, _ = timed(S.MATCH_AUTO)                      # what sfm_match runs by default; the record's headline figures
    flops = 2.0 * n * n * 128                             # ALGORITHMIC flops of the all-pairs scores
    ok = float((idx.cpu().numpy() == perm).mean())
    row = {"n": n, "ms": ms, "kernel": NAMES[ran], "tflops": flops / ms / 1e9, "frac_of_157.3": flops / ms / 1e9 / 157.3, "perm_recovered": ok,
           "ms_exact": ms_exact, "ms_prefilter": ms_pf,
           "prefilter_equals_exact_bitwise": bool(all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(res_exact, res_pf)))}
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
