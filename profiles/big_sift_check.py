import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cuda_sfm_amd as S, oracle as O
from cuda_sfm_amd import synth
from helpers import same_bits
w, h = 4096, 3072
t = time.time(); img = synth.image(w, h, seed=3, blobs=6000); print("gen", time.time() - t)
dev = torch.device("cuda:0"); ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
p = w
d_img = torch.from_numpy(img).to(dev)
maxp = 262144
d_sift = torch.zeros((maxp, 576), dtype=torch.uint8, device=dev)
for _ in range(2):
    n, stored = ctx.extract_sift(d_sift, maxp, d_img, w, h, p, 6, 1.0, 3.0)
torch.cuda.synchronize(); t = time.time()
for _ in range(5):
    n, stored = ctx.extract_sift(d_sift, maxp, d_img, w, h, p, 6, 1.0, 3.0)
torch.cuda.synchronize(); print("gpu ms", (time.time() - t) / 5 * 1e3, n, stored)
rec = d_sift.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)
t = time.time(); opts, on, ost = O.extract_sift(img, 6, 1.0, 3.0, max_pts=maxp); print("oracle s", time.time() - t, on, ost)
assert (n, stored) == (on, ost)
for f in ("xpos", "ypos", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data"):
    assert same_bits(rec[f][:stored], opts[f][:stored]), f
print("big image parity ok")
