"""Contexts are independent (SURVEY 8b "Threading": one context per host thread, each on its own HIP stream): two
host threads drive their own context concurrently through RANSAC, the matcher and SIFT extraction; every result
must equal what the same context configuration produces alone (which the other GPU tests pin to the oracle)."""
import threading

import numpy as np
import pytest

import cuda_sfm_amd as S
from cuda_sfm_amd_synth import synth
from helpers import same_bits, to_dev

pytestmark = pytest.mark.gpu


def _job(torch, dev, stream, tid, out, rounds):
    with torch.cuda.stream(stream):
        ctx = S.Context(dev.index or 0, stream.cuda_stream)
        n, H = 1500 + 700 * tid, 3000 + 1000 * tid
        scene = synth.two_view_scene(n, seed=40 + tid)
        d1, d2, _ = synth.descriptors(900 + 300 * tid, seed=60 + tid)
        w, h = 320 + 64 * tid, 240
        img = synth.image(w, h, seed=70 + tid, blobs=120)
        pitch = (w + 127) // 128 * 128
        pad = np.zeros((h, pitch), np.float32); pad[:, :w] = img
        d_sift_pts = to_dev(torch, dev, scene["sift"])
        t1, t2 = to_dev(torch, dev, d1), to_dev(torch, dev, d2)
        d_img = torch.from_numpy(pad).to(dev)
        res = []
        for _ in range(rounds):
            pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
            pair.fillXU(d_sift_pts)
            pair.estimateE(S.default_params(n, num_hypotheses=H, seed=5 + tid, kernel=S.KERNEL_SPLIT))
            best = torch.empty(len(d1), dtype=torch.float32, device=dev); sec = torch.empty_like(best)
            idx = torch.empty(len(d1), dtype=torch.int32, device=dev)
            ctx.match_soa(t1, len(d1), 128, t2, len(d2), 128, best, sec, idx)
            d_out = torch.zeros((4096, 576), dtype=torch.uint8, device=dev)
            npts, stored = ctx.extract_sift(d_out, 4096, d_img, w, h, pitch, num_octaves=4, init_blur=1.0, thresh=2.5)
            stream.synchronize()
            res.append((pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy(), pair.get_inlier_counts(H).copy(),
                        idx.cpu().numpy(), best.cpu().numpy(), npts, stored, d_out.cpu().numpy()[:stored].copy()))
            pair.close()
        ctx.close()
        out[tid] = res


def test_two_threads_two_contexts(gpu):
    torch, dev, _ = gpu
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    alone = {}
    for tid in (0, 1):                                   # baseline: each job alone
        _job(torch, dev, streams[tid], tid, alone, 1)
    together = {}
    threads = [threading.Thread(target=_job, args=(torch, dev, streams[tid], tid, together, 6)) for tid in (0, 1)]
    for t in threads: t.start()
    for t in threads: t.join(300)
    assert sorted(together) == [0, 1], "a worker thread died"
    for tid in (0, 1):
        key, E, mask, counts, idx, best, npts, stored, rec = alone[tid][0]
        assert npts > 30 and mask.sum() > 100
        for r in together[tid]:
            assert r[0] == key and same_bits(r[1], E) and np.array_equal(r[2], mask) and np.array_equal(r[3], counts)
            assert np.array_equal(r[4], idx) and same_bits(r[5], best)
            assert (r[6], r[7]) == (npts, stored) and np.array_equal(r[8], rec)
