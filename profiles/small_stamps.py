"""fill_xu / finalize / pose chain of the dino pair: wall-clock stamps (100 MHz) inside the kernels of a debug build
(recipe in profiles/r04_small_kernel_stamps.txt)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import cuda_sfm_amd as S
from helpers import read_pnm_grey, dino_frame, DINO_K, DINO_KINV, DINO_SIFT
dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
lib = ctypes.CDLL(os.path.join(os.environ["SFM_AMD_LIB_DIR"], "libsfm_amd.so"))
views = [read_pnm_grey(dino_frame(k)) for k in range(2)]
h, w = views[0].shape; pitch = (w + 127) // 128 * 128
def extract(img):
    pad = np.zeros((h, pitch), np.float32); pad[:, :w] = img
    d = torch.zeros((32768, 576), dtype=torch.uint8, device=dev)
    n, _ = ctx.extract_sift(d, 32768, torch.from_numpy(pad).to(dev), w, h, pitch, **DINO_SIFT)
    return d, n
(s1, n1), (s2, n2) = extract(views[0]), extract(views[1])
pair = S.ImagePair(ctx, DINO_K, DINO_KINV, 2, n1)
p = S.default_params(n1, num_hypotheses=1024)
for _ in range(30):
    ctx.match(s1, n1, s2, n2); pair.fillXU(s1); pair.estimateE(p); pair.pose_chain()
torch.cuda.synchronize()
buf = np.zeros(3 * 1024, np.uint64)
lib.sfm_dbg_small_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
fin = np.zeros(16, np.uint64)
lib.sfm_dbg_fin_stamps(fin.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(fin.nbytes))
def show(title, st, names):
    st = st[st[:, 0] > 0].astype(np.int64)
    t0 = st[:, 0].min()
    print(title, f"({len(st)} blocks)")
    for k, nm in enumerate(names):
        col = (st[:, k] - t0) * 0.01
        print(f"  {nm:28s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us")
show("fill_xu_kernel", buf[:1024].reshape(64, 16), ["entry", "records loaded, X computed", "stores + atomic issued", "all acknowledged"])
show("pose_chain_reference_kernel", buf[1024:2048].reshape(64, 16), ["entry", "E loaded", "candidates (svd3) done", "inverse done", "triangulated", "after barrier", "stored"])
f = fin.astype(np.int64)
print("ransac_finalize_block")
for k, nm in enumerate(["entry", "key read", "E in LDS (barrier)", "mask + count loop done", "best written"]):
    print(f"  {nm:28s} {(f[k] - f[0]) * 0.01:6.2f} us")
