"""dino pair (BASELINE configs[1]): 60 end-to-end iterations (match -> fillXU -> estimateE -> pose chain) for a kernel trace.
Run under `rocprofv3 --kernel-trace` by c1_timeline.sh, which turns the trace into durations and gaps per kernel of one iteration."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
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
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = S.default_params(n1, num_hypotheses=H)
def e2e():
    ctx.match(s1, n1, s2, n2); pair.fillXU(s1); pair.estimateE(p); pair.pose_chain()
for _ in range(30): e2e()
ctx.synchronize(); ctx.timer_start()
for _ in range(60): e2e()
print("e2e_us %.2f" % (1e3 * ctx.timer_stop() / 60), "n1", n1, "n2", n2, "H", H, "best", pair.get_best())
