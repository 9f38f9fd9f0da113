import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import torch
import cuda_sfm_amd_ab as S
from cuda_sfm_amd_ab import synth
import oracle as O
dev = torch.device("cuda", 0); ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
for n, H, seed in ((4096, 1 << 18, 3), (5000, 65536, 4), (16384, 65536, 5), (700, 20000, 6)):
    scene = synth.two_view_scene(n, seed=seed)
    d = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n); pair.fillXU(d)
    p = S.default_params(n, num_hypotheses=H, seed=seed, kernel=S.KERNEL_PREFILTER); p.reserved[1] = 9
    pair.estimateE(p)
    _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
    key, oc, _ = O.ransac_range_fast(X0, X1, 0, H, p.threshold, 0, seed=seed)
    c = pair.get_inlier_counts(H)
    print(n, H, "lds", pair.last_launch()["lds_bytes"], "bad", int((c != oc).sum()), pair.get_key() == key)
