"""Matcher timings at a few sizes (AUTO), for same-box A/B of two builds (SFM_AMD_LIB_DIR)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
out = {}
for n in (1200, 1500, 1800, 2048, 2155, 2300, 2500, 3000, 4000, 5500, 16384):
    d1, d2, perm = synth.descriptors(n)
    t1, t2 = torch.from_numpy(d1).to(dev), torch.from_numpy(d2).to(dev)
    best = torch.empty(n, dtype=torch.float32, device=dev); sec = torch.empty_like(best); idx = torch.empty(n, dtype=torch.int32, device=dev)
    for kern in ((S.MATCH_AUTO, S.MATCH_EXACT) if n <= 5500 else (S.MATCH_AUTO,)):
        ctx.set_match_kernel(kern)
        for _ in range(20): ctx.match_soa(t2, n, 128, t1, n, 128, best, sec, idx)
        torch.cuda.synchronize(); ctx.timer_start()
        reps = 200 if n <= 5500 else 40
        for _ in range(reps): ctx.match_soa(t2, n, 128, t1, n, 128, best, sec, idx)
        out[f"{n}{'' if kern == S.MATCH_AUTO else '_exact'}"] = round(1e3 * ctx.timer_stop() / reps, 2)
    assert (idx.cpu().numpy() == perm).mean() > 0.99
print(json.dumps({"lib": os.environ.get("SFM_AMD_LIB_DIR", "new")[-12:], "us": out}))
