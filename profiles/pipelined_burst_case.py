"""Replay of a burst the stateful fuzz round flagged (tests/fuzz_gpu.py, seed 502): two pipelined estimateE calls on a fresh pair -- a large SPLIT call
(Jacobi solver) on slot 0 and a small AUTO (fused kernel) call on slot 1 -- the result must be the second call's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cuda_sfm_amd as S
from cuda_sfm_amd import synth
import oracle as O
from helpers import same_bits

dev = torch.device("cuda:0")
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
n = 4735
scene = synth.two_view_scene(n, seed=522527723 & 0x7FFFFFFF)
d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
_, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
B, Sm = (20309, 3, 6.18e-7, 58246), (546, 7, 1.27e-5, 50380)
def call(c, kernel, sweeps=None):
    return (c[0], kernel, c[1] if sweeps is None else sweeps, c[2], c[3])
variants = {
    "fused big sweeps 3, fused small sweeps 7": [call(B, S.KERNEL_FUSED), call(Sm, S.KERNEL_FUSED)],
    "fused big sweeps 0, fused small sweeps 0": [call(B, S.KERNEL_FUSED, 0), call(Sm, S.KERNEL_FUSED, 0)],
    "fused big sweeps 3, fused small sweeps 0": [call(B, S.KERNEL_FUSED), call(Sm, S.KERNEL_FUSED, 0)],
    "fused big sweeps 0, fused small sweeps 7": [call(B, S.KERNEL_FUSED, 0), call(Sm, S.KERNEL_FUSED)],
    "fused big sweeps 3, split small sweeps 7": [call(B, S.KERNEL_FUSED), call(Sm, S.KERNEL_SPLIT)],
    "split big sweeps 3, fused small sweeps 7": [call(B, S.KERNEL_SPLIT), call(Sm, S.KERNEL_FUSED)],
    "fused big, FLUSH, fused small": [call(B, S.KERNEL_FUSED), "flush", call(Sm, S.KERNEL_FUSED)],
    "fused small, fused small (same)": [call(Sm, S.KERNEL_FUSED), call(Sm, S.KERNEL_FUSED)],
}
for name, burst in variants.items():
    H, k, sw, thr, seed = burst[-1]
    key, _, oE = O.ransac_range(X0, X1, 0, H, np.float32(thr), sw, seed=seed, want_E=True)
    ocnt, ohyp = O.unpack_key(key)
    omask = O.count_inliers(oE[ohyp], X0, X1, np.float32(thr))[1]
    bad, exc, kinds = 0, 0, {}
    for rep in range(200):
        pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
        pair.fillXU(d_sift)
        for c in burst:
            if c == "flush":
                pair.flush(); torch.cuda.synchronize()
                continue
            (h, kern, sweeps, t, sd) = c
            pair.estimateE_pipelined(S.default_params(n, num_hypotheses=h, seed=sd, kernel=kern, jacobi_sweeps=sweeps, threshold=t))
        try:
            got = pair.get_best()
            ok = got == (ohyp, ocnt) and same_bits(pair.get_E(), oE[ohyp].reshape(3, 3)) and np.array_equal(pair.get_inlier_mask(), omask)
            if not ok:
                bad += 1; kinds[str(got)] = kinds.get(str(got), 0) + 1
        except S.SfmError as e:
            exc += 1
        pair.close()
    print(f"{name}: wrong {bad}, exceptions {exc} of 200; want {(ohyp, ocnt)}; wrong winners {kinds}", flush=True)
