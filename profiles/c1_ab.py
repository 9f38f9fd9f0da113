"""dino pair (BASELINE configs[1]) stage timings, for same-box A/B of two builds (SFM_AMD_LIB_DIR)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cuda_sfm_amd as S
from helpers import read_pnm_grey, dino_frame, DINO_K, DINO_KINV, DINO_SIFT
dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
views = [read_pnm_grey(dino_frame(k)) for k in range(2)]
h, w = views[0].shape; pitch = (w + 127) // 128 * 128
def extract(img):
    pad = np.zeros((h, pitch), np.float32); pad[:, :w] = img
    d = torch.zeros((32768, 576), dtype=torch.uint8, device=dev)
    n, _ = ctx.extract_sift(d, 32768, torch.from_numpy(pad).to(dev), w, h, pitch, **DINO_SIFT)
    return d, n
(s1, n1), (s2, n2) = extract(views[0]), extract(views[1])
pair = S.ImagePair(ctx, DINO_K, DINO_KINV, 2, n1)
def timed(fn, reps=200):
    for _ in range(20): fn()
    ctx.synchronize(); ctx.timer_start()
    for _ in range(reps): fn()
    return round(1e3 * ctx.timer_stop() / reps, 2)
out = {}
for H in (n1 // 8, 1024):
    p = S.default_params(n1, num_hypotheses=H)
    def e2e():
        ctx.match(s1, n1, s2, n2); pair.fillXU(s1); pair.estimateE(p); pair.pose_chain()
    ctx.match(s1, n1, s2, n2); pair.fillXU(s1)
    out[f"H{H}"] = {"estimateE_us": timed(lambda: pair.estimateE(p)), "match_us": timed(lambda: ctx.match(s1, n1, s2, n2)), "e2e_us": timed(e2e), "best": pair.get_best()}
print(json.dumps({"lib": os.environ.get("SFM_AMD_LIB_DIR", "new"), **out}))
