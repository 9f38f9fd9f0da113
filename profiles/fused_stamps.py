"""Where the time of one fused estimateE launch (one hypothesis per wavefront) goes on the dino pair: wall-clock stamps (100 MHz)
by thread 0 of every block of a debug build (-DSFM_FUSED_STAMPS; recipe in profiles/r04_fused_stamps.txt)."""
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
ctx.match(s1, n1, s2, n2); pair.fillXU(s1)
names = ["entry", "schedule built", "sample gathered", "(copied to regs)", "Householder done", "normalize_E done", "before staging", "tile staged", "scored", "key out"]
for H in (269, 1024):
    p = S.default_params(n1, num_hypotheses=H)
    for _ in range(30): pair.estimateE(p)
    torch.cuda.synchronize()
    buf = np.zeros(16 * 1024, np.uint64)
    lib.sfm_dbg_fused_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    st = buf.reshape(1024, 16).astype(np.int64)
    nb = (H + 3) // 4 if (H + 3) // 4 < 256 else 256
    st = st[:nb]
    t0 = st[:, 0].min()
    rel = (st - t0) * 0.01
    print(f"H = {H}: {nb} blocks of 4 wavefronts")
    for k, nm in enumerate(names):
        col = rel[:, k]
        print(f"  {nm:20s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us")
