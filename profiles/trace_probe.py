"""When do the blocks and wavefronts of ransac_score_prefilter start and finish inside ONE launch (sfm_ransac_last_trace)?
Prints, per configuration: kernel span, block lifetimes (percentiles), how far the wavefront ends of a block are spread,
per-XCC mean lifetime, and the mean number of wavefronts still running per CU over the span (what the VALU can overlap).
Run on the GPU box: python profiles/trace_probe.py [hyps ...]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cuda_sfm_amd_ab as S            # the lab-bench flavour (make ab): switches, probes, traces
from cuda_sfm_amd_ab import synth

dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
cases = [(4096, int(h)) for h in sys.argv[1:]] or [(4096, 1 << 20), (4096, 131072), (16384, 1 << 20)]
for n, H in cases:
    scene = synth.two_view_scene(n)
    d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.fillXU(d_sift)
    p = S.default_params(n, num_hypotheses=H, kernel=S.KERNEL_PREFILTER)
    for _ in range(10):
        pair.ransac_score(p)
    torch.cuda.synchronize()
    t = pair.last_trace().astype(np.float64)
    us = 0.01
    start, wave_end = t[:, 0] * us, t[:, 4:20] * us
    t0 = start.min()
    start -= t0; wave_end -= t0
    blk_end = wave_end.max(axis=1)
    life = blk_end - start
    span = blk_end.max()
    xcc = (t[:, 2].astype(np.uint64) >> np.uint64(32)).astype(int) & 0xF
    spread = wave_end.max(axis=1) - wave_end.min(axis=1)
    # mean number of live wavefronts (of 16 per block) over the kernel span
    live = (wave_end - start[:, None]).sum() / (span * len(t))
    out = {"matches": n, "hypotheses": H, "blocks": len(t), "span_us": round(span, 1), "last_start_us": round(float(start.max()), 1),
           "block_lifetime_us_p0_10_50_90_100": [round(float(x), 1) for x in np.percentile(life, [0, 10, 50, 90, 100])],
           "block_end_us_p10_50_90": [round(float(x), 1) for x in np.percentile(blk_end, [10, 50, 90])],
           "wave_end_spread_in_block_us_p50_90_100": [round(float(x), 1) for x in np.percentile(spread, [50, 90, 100])],
           "mean_live_waves_per_block_over_span": round(float(live), 2),
           "lifetime_by_xcc_us": {int(x): round(float(life[xcc == x].mean()), 1) for x in sorted(set(xcc))},
           "blocks_by_xcc": {int(x): int((xcc == x).sum()) for x in sorted(set(xcc))}}
    print(json.dumps(out), flush=True)
    pair.close()
