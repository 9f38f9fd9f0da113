"""Debug build only (a scoring kernel that stamps every block's start and end into a __device__ array and exports
sfm_debug_block_times): when do the blocks of ransac_score_prefilter start and finish inside one launch?"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cuda_sfm_amd_ab as S            # the lab-bench flavour (make ab): switches, probes, traces
from cuda_sfm_amd_ab import synth
dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
n = 4096
scene = synth.two_view_scene(n)
d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
pair.fillXU(d_sift)
lib = S._lib
for H, cols in ((131072, 0), (131072, 64), (1048576, 0)):
    p = S.default_params(n, num_hypotheses=H, seed=3, kernel=S.KERNEL_PREFILTER)
    p.reserved[2] = cols
    for _ in range(10):
        pair.estimateE(p)
    torch.cuda.synchronize()
    pair.estimateE(p)
    torch.cuda.synchronize()
    g = pair.last_launch()["grid"]
    t = (C.c_uint64 * (2 * g))()
    lib.sfm_debug_block_times(t, 2 * g)
    a = np.array(t, dtype=np.uint64).reshape(g, 2).astype(np.float64) / 100.0     # us (100 MHz)
    t0 = a[:, 0].min()
    st, en = a[:, 0] - t0, a[:, 1] - t0
    print(json.dumps({"hypotheses": H, "grid": g, "last_start_us": round(float(st.max()), 1), "start_percentiles_us": [round(float(x), 1) for x in np.percentile(st, [10, 50, 90])],
                      "end_percentiles_us": [round(float(x), 1) for x in np.percentile(en, [10, 50, 90, 100])], "lifetime_median_us": round(float(np.median(en - st)), 1),
                      "lifetime_p90_us": round(float(np.percentile(en - st, 90)), 1)}), flush=True)
