"""Lane-solve kernel A/B: sampled points gathered as scattered dwords (reserved[0] = 4) or as 16-byte records (default), packed\n(two hypotheses per lane) or scalar (reserved[0] = 3) Householder kernel; candidates must be bit-identical."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import cuda_sfm_amd_ab as S            # the lab-bench flavour (make ab): switches, probes, traces
from cuda_sfm_amd_ab import synth
dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
n = 4096
scene = synth.two_view_scene(n)
pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
pair.fillXU(torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev))
for H in (4096, 16384, 65536, 131072, 262144, 1048576):
    res = {}
    for r in (4, 0, 3):
        p = S.default_params(n, num_hypotheses=H, seed=5)
        p.reserved[0] = r
        for _ in range(5):
            pair.estimateE(p)
        ctx.synchronize()
        ctx.kernel_timing(True)
        for _ in range(20):
            pair.estimateE(p)
        solve_ms, score_ms, calls = ctx.kernel_timing_read()
        ctx.kernel_timing(False)
        ctx.timer_start()
        for _ in range(50):
            pair.estimateE(p)
        step = ctx.timer_stop() / 50
        res[r] = (solve_ms / calls, score_ms / calls, step, pair.get_E_candidates(H).copy(), pair.get_key())
    same = all(np.array_equal(res[4][3].view(np.uint32), res[r][3].view(np.uint32)) and res[4][4] == res[r][4] for r in (0, 3))
    print(json.dumps({"hypotheses": H, "packed_scattered_solve_us": round(1e3 * res[4][0], 1), "packed_records_solve_us": round(1e3 * res[0][0], 1), "scalar_records_solve_us": round(1e3 * res[3][0], 1),
                      "step_us": [round(1e3 * res[r][2], 1) for r in (4, 0, 3)], "candidates_bit_identical": bool(same)}), flush=True)
