"""What a call on a FRESH point set costs (fillXU + estimateE, the reference's one-shot use: sfm.cu:409-431 runs each once per image pair) under
the three ways of choosing the pre-filter's form, measured alternately on one box (five rounds of 20 iterations each, median and minimum):
reserved[3] = 6 per-hypothesis operands for every call (no ordered copy, no tile boxes), 7 per-tile operands for every call (the first builds
the ordered copy), 0 the product's choice (first call per hypothesis below 2^33 pairs).  Lab-bench library."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import cuda_sfm_amd_ab as S
from cuda_sfm_amd_ab import synth
dev = torch.device("cuda", 0)
ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
NAMES = {0: "the product's choice", 6: "per-hypothesis operands always", 7: "per-tile operands always"}
for n, H in ((4096, 1 << 20), (16384, 65536), (4096, 131072), (16384, 1 << 20), (4096, 1 << 21), (1024, 32768)):
    scene = synth.two_view_scene(n)
    d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
    pairs, params = {}, {}
    for r3 in NAMES:
        pairs[r3] = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
        params[r3] = S.default_params(n, num_hypotheses=H)
        params[r3].reserved[3] = r3
        for _ in range(3):
            pairs[r3].fillXU(d_sift); pairs[r3].estimateE(params[r3])
    torch.cuda.synchronize()
    fresh = {r3: [] for r3 in NAMES}
    warm = {r3: [] for r3 in NAMES}
    reps = 20
    for rnd in range(5):
        for r3 in NAMES:
            pair, p = pairs[r3], params[r3]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                pair.fillXU(d_sift); pair.estimateE(p)          # a fresh point set every time
            torch.cuda.synchronize()
            fresh[r3].append((time.perf_counter() - t0) / reps)
            pair.estimateE(p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                pair.estimateE(p)
            torch.cuda.synchronize()
            warm[r3].append((time.perf_counter() - t0) / reps)
    for r3 in NAMES:
        print(f"n {n} H {H} {NAMES[r3]:32s}: fillXU + estimateE on a fresh point set {1e3 * np.median(fresh[r3]):.4f} ms (min {1e3 * min(fresh[r3]):.4f}), "
              f"estimateE again on the same points {1e3 * np.median(warm[r3]):.4f} ms (min {1e3 * min(warm[r3]):.4f})", flush=True)
