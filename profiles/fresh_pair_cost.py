import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import cuda_sfm_amd_ab as S            # the lab-bench flavour: reserved[3] = 6 selects per-hypothesis records (no ordered copy, no tile boxes)
from cuda_sfm_amd_ab import synth
dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
for n, H, r3 in ((4096, 1 << 20, 0), (4096, 1 << 20, 6), (16384, 65536, 0), (16384, 65536, 6), (4096, 131072, 0), (4096, 131072, 6)):
    scene = synth.two_view_scene(n)
    d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    p = S.default_params(n, num_hypotheses=H)
    p.reserved[3] = r3
    for _ in range(3):
        pair.fillXU(d_sift); pair.estimateE(p)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        pair.fillXU(d_sift); pair.estimateE(p)          # a fresh point set every time: cell table, ordered copy, tile boxes rebuilt
    torch.cuda.synchronize()
    fresh = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        pair.estimateE(p)
    torch.cuda.synchronize()
    warm = (time.perf_counter() - t0) / reps
    print(f"n {n} H {H} {'per-tile rule' if r3 == 0 else 'per-hypothesis records'}: fillXU + estimateE on a fresh point set {1e3 * fresh:.4f} ms, estimateE alone {1e3 * warm:.4f} ms, the once-per-fillXU part (fillXU, cells, ordering, boxes) {1e3 * (fresh - warm):.4f} ms", flush=True)
