"""Time-budgeted randomised parity run on the GPU.

    python tests/fuzz_gpu.py [seconds] [seed] [--lib product|ab]

Every round draws a random configuration of one of the product paths (RANSAC in all kernel families and both
null-vector solvers, with ordinary, degenerate, generic-z and huge-coordinate point sets; the matcher; the
homography search; SIFT extraction), runs it through the C ABI and compares with the CPU oracle bit for bit.
Prints one JSON line with the number of rounds per path and every mismatch (configuration included).
--lib product (default) loads libsfm_amd.so, what ships; --lib ab the lab-bench flavour, which adds the recorded A/B kernel
variants to the draw.  tests/test_gpu_fuzz_slice.py runs a bounded slice of the same rounds (product library, seed from the
committed tests/fuzz_seed.txt) under pytest -m gpu, i.e. on the driver's box.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def make_rounds(S, torch, dev, ctx, rng):
    """[(name, round function -> (ok, cfg), probability)] for library flavour S (cuda_sfm_amd or cuda_sfm_amd_ab)."""
    import importlib
    synth = importlib.import_module(S.__name__ + ".synth")
    import oracle as O
    from helpers import same_bits, to_dev
    kernels = [S.KERNEL_AUTO, S.KERNEL_SPLIT, S.KERNEL_FUSED, S.KERNEL_PREFILTER, S.KERNEL_PREFILTER]
    if hasattr(S, "KERNEL_MFMA"):
        kernels.append(S.KERNEL_MFMA)

    def ransac_round():
        n = int(rng.choice([rng.integers(8, 200), rng.integers(200, 3000), rng.integers(3000, 9000)]))
        H = int(rng.choice([rng.integers(1, 64), rng.integers(64, 1500), rng.integers(1500, 6000), rng.integers(6000, 24000)]))
        kernel = int(rng.choice(kernels))
        sweeps = int(rng.choice([0, 0, 7, 3]))
        thr = float(np.float32(10.0 ** rng.uniform(-9, -2)))
        flavour = str(rng.choice(["plain", "plain", "clean", "dup", "epipole", "genericz", "huge", "nan"]))
        if kernel == S.KERNEL_PREFILTER:              # the matrix-core pre-filter needs fillXU points (z == 1): wide / narrow
            flavour = str(rng.choice(["plain", "clean", "dup", "nan", "forward", "wide"]))       # fields of view, forward motion (epipole in the image)
            H = int(rng.choice([H, rng.integers(64, 3000)]))
        sseed = int(rng.integers(1, 1 << 30))
        scene = synth.two_view_scene(n, seed=sseed, noise_px=float(rng.choice([0.0, 0.3, 2.0])),
                                     outlier_frac=float(rng.choice([0.0, 0.3, 0.8])) if flavour != "clean" else 0.0,
                                     focal=float(rng.choice([150.0, 600.0, 9000.0])) if flavour == "wide" else 2360.0)
        if flavour == "forward":                      # matches that move radially from the principal point: epipoles inside the image,
            c = np.array([360.0, 288.0]); s1 = scene["sift"]          # some correspondences exactly on it (zero divisors in the residual)
            d = np.stack([s1["xpos"], s1["ypos"]], 1) - c
            s1["match_xpos"], s1["match_ypos"] = (c + 1.07 * d).T.astype(np.float32)
            on = rng.integers(0, n, max(1, n // 40))
            s1["xpos"][on] = 360.0; s1["ypos"][on] = 288.0; s1["match_xpos"][on] = 360.0; s1["match_ypos"][on] = 288.0
        cfg = dict(path="ransac", n=n, H=H, kernel=kernel, sweeps=sweeps, thr=thr, flavour=flavour, scene_seed=sseed)
        sift = scene["sift"]
        if flavour == "dup":                          # repeated correspondences -> rank-deficient samples
            k = int(rng.integers(1, max(2, n // 2)))
            src = rng.integers(0, n, k); dst = rng.integers(0, n, k)
            for f in ("xpos", "ypos", "match_xpos", "match_ypos"):
                sift[f][dst] = sift[f][src]
        if flavour == "nan":
            sift["xpos"][rng.integers(0, n, max(1, n // 50))] = np.nan
        _, _, X0, X1 = O.fill_xu(sift, scene["Kinv"])
        pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
        if flavour in ("genericz", "huge", "epipole"):
            X0 = X0.copy(); X1 = X1.copy()
            if flavour == "genericz":
                X0 *= (0.5 + synth.uniform01(sseed, n)).astype(np.float32); X1 *= (2.0 - synth.uniform01(sseed + 1, n)).astype(np.float32)
            elif flavour == "huge":
                sc = np.float32(10.0 ** rng.uniform(2, 9)); X0 *= sc; X1 *= sc
                cfg["scale"] = float(sc)
            else:                                     # many points exactly on one spot of image 2 (a likely epipole of bad hypotheses)
                idx = rng.integers(0, n, max(1, n // 10)); X1[0, idx] = X1[0, idx[0]]; X1[1, idx] = X1[1, idx[0]]
            X0 = np.ascontiguousarray(X0, np.float32); X1 = np.ascontiguousarray(X1, np.float32)
            pair.set_points(to_dev(torch, dev, X0), to_dev(torch, dev, X1))
        else:
            pair.fillXU(to_dev(torch, dev, sift))
        p = S.default_params(n, num_hypotheses=H, seed=sseed & 0xFFFF, kernel=kernel, jacobi_sweeps=sweeps, threshold=thr)
        pair.estimateE(p)
        if kernel == S.KERNEL_PREFILTER and rng.random() < 0.6:       # a second call on the same points: per-tile operands over the ordered copy
            pair.estimateE(p)
            cfg["calls"] = 2
        key, ocounts, oE = O.ransac_range(X0, X1, 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed, want_E=True)
        ocnt, ohyp = O.unpack_key(key)
        _, omask = O.count_inliers(oE[ohyp], X0, X1, p.threshold)
        checks = {"counts": np.array_equal(pair.get_inlier_counts(H), ocounts), "key": pair.get_key() == key,
                  "candidates": same_bits(pair.get_E_candidates(H), oE), "mask": np.array_equal(pair.get_inlier_mask(), omask),
                  "E": same_bits(pair.get_E(), oE[ohyp].reshape(3, 3))}
        ok = all(checks.values())
        if not ok:                                    # which comparison failed, and what ran
            cfg["failed"] = [k for k, v in checks.items() if not v]
            cfg["launch"] = pair.last_launch()
            if os.environ.get("FUZZ_DUMP"):           # the whole case, for a closer look (profiles/fuzz_case.py)
                np.savez(os.path.join(os.environ["FUZZ_DUMP"], f"fuzz_fail_{n}_{H}.npz"), sift=sift.view(np.uint8), K=scene["K"], Kinv=scene["Kinv"], thr=p.threshold,
                         seed=p.seed, sweeps=sweeps, H=H, n=n, kernel=kernel, counts=pair.get_inlier_counts(H), ocounts=ocounts, calls=cfg.get("calls", 1))
        return ok, cfg

    def calls_round():
        """ONE pair through a random sequence of calls -- what depends on the pair's state: the fillXU epoch (bound, cell table, whole-view boxes,
        the ordered copy and its tile boxes), which form of the pre-filter a call runs (first call after a fillXU per hypothesis, later ones per
        tile), the two slots of pipelined calls, records written by the lane-solve kernel / by the stand-alone kernel (Jacobi solver, supplied
        candidates).  Every step is checked: counts, key, E and mask (pipelined bursts: winner, E, mask)."""
        n = int(rng.choice([rng.integers(300, 1500), rng.integers(1500, 6000)]))
        sseed = int(rng.integers(1, 1 << 30))
        cfg = dict(path="calls", n=n, scene_seed=sseed, steps=[])
        # the context: the shared one (torch's default stream, i.e. the null stream) or one of its own on a non-blocking stream -- what is
        # ordered by accident on the null stream is not on the other
        own = bool(rng.random() < 0.4)
        cfg["own_stream"] = own
        c = ctx
        if own:
            c = S.Context(0)
            c.own_stream()
        pair = S.ImagePair(c, np.eye(3, dtype=np.float32), np.eye(3, dtype=np.float32), 2, n)
        state = {"ctx": c}

        def refill():
            sc = synth.two_view_scene(n, seed=int(rng.integers(1, 1 << 30)), noise_px=float(rng.choice([0.0, 0.3, 2.0])), outlier_frac=float(rng.choice([0.0, 0.3, 0.8])),
                                      focal=float(rng.choice([600.0, 2360.0, 2360.0, 9000.0])))
            # (the pair keeps its K: X = U with K = I would leave pixel coordinates; use the scene's normalised coordinates as "pixels" of an identity camera)
            _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
            sift = np.zeros(n, synth.SIFT_DTYPE)
            sift["xpos"], sift["ypos"], sift["match_xpos"], sift["match_ypos"] = X0[0, :n], X0[1, :n], X1[0, :n], X1[1, :n]
            pair.fillXU(to_dev(torch, dev, sift))
            _, _, state["X0"], state["X1"] = O.fill_xu(sift, np.eye(3, dtype=np.float32))

        def params():
            H = int(rng.choice([rng.integers(64, 1500), rng.integers(1500, 8000), rng.integers(16384, 24000)]))
            kernel = int(rng.choice([S.KERNEL_AUTO, S.KERNEL_SPLIT, S.KERNEL_PREFILTER, S.KERNEL_PREFILTER, S.KERNEL_PREFILTER]))
            return H, S.default_params(n, num_hypotheses=H, seed=int(rng.integers(0, 1 << 16)), kernel=kernel, jacobi_sweeps=int(rng.choice([0, 0, 0, 7, 3])),
                                       threshold=float(np.float32(10.0 ** rng.uniform(-7, -3))))

        def want(H, p):
            key, ocounts, oE = O.ransac_range(state["X0"], state["X1"], 0, H, p.threshold, p.jacobi_sweeps, seed=p.seed, want_E=True)
            ocnt, ohyp = O.unpack_key(key)
            return key, ocounts, oE[ohyp].reshape(3, 3), O.count_inliers(oE[ohyp], state["X0"], state["X1"], p.threshold)[1], (ohyp, ocnt)

        refill()
        nsteps = int(rng.integers(2, 7))
        try:
            return calls_steps(pair, state, cfg, nsteps, refill, params, want)
        except Exception as e:                        # keep the sequence that led to it
            cfg["exception"] = repr(e)
            cfg["launch"] = pair.last_launch()
            return False, cfg

    def calls_steps(pair, state, cfg, nsteps, refill, params, want):
        n = cfg["n"]
        for step in range(nsteps):
            op = str(rng.choice(["estimate", "estimate", "estimate", "pipelined", "candidates", "shards", "refill", "pose"]))
            if op == "refill":
                refill()
                state["have_E"] = False
                cfg["steps"].append(op)
                continue
            if op == "pose":                          # the pose stages on whatever E the last step left (they flush pending pipelined steps themselves)
                if not state.get("have_E"):
                    continue
                mode = int(rng.choice([S.POSE_REFERENCE, S.POSE_CORRECT]))
                chain = bool(rng.random() < 0.5)
                cfg["steps"].append(dict(op=op, mode=mode, chain=chain))
                if chain:
                    pair.pose_chain(mode)
                else:
                    pair.computePosecandidates(mode); pair.choosePose(mode); pair.linear_triangulation(mode)
                oP = O.pose_candidates(pair.get_E(), mode)
                oind, oPinv, _, _ = O.choose_pose(state["X0"], state["X1"], oP, mode, sweeps=8)
                try:
                    ind = pair.get_pose_index()
                except S.SfmError as e:               # a singular chosen candidate is reported
                    if e.code != S.E_SINGULAR: raise
                    continue
                ok = ind == oind and same_bits(pair.get_pose_candidates(), oP) and same_bits(pair.get_pose_inverses(), oPinv)
                ok = ok and same_bits(pair.get_points(), O.triangulate(state["X0"], state["X1"], oPinv[oind] if mode == S.POSE_REFERENCE else oP[oind], sweeps=8))
                if not ok:
                    cfg["failed_step"] = step
                    pair.close()
                    return False, cfg
                continue
            H, p = params()
            cfg["steps"].append(dict(op=op, H=H, kernel=p.kernel, sweeps=p.jacobi_sweeps, thr=p.threshold, seed=p.seed))
            if op == "estimate":
                pair.estimateE(p)
                key, ocounts, oEb, omask, best = want(H, p)
                ok = np.array_equal(pair.get_inlier_counts(H), ocounts) and pair.get_key() == key and same_bits(pair.get_E(), oEb) and np.array_equal(pair.get_inlier_mask(), omask)
            elif op == "pipelined":
                burst = [(H, p)] + [params() for _ in range(int(rng.integers(0, 3)))]
                cfg["steps"][-1]["burst"] = [dict(H=h, kernel=q.kernel, sweeps=q.jacobi_sweeps, thr=q.threshold, seed=q.seed) for h, q in burst]
                for _, q in burst:
                    pair.estimateE_pipelined(q)
                key, ocounts, oEb, omask, best = want(*burst[-1])
                ok = pair.get_best() == best and same_bits(pair.get_E(), oEb) and np.array_equal(pair.get_inlier_mask(), omask)
            elif op == "shards":                      # what G ranks do, one after the other: score a shard into a caller's key, reduce, finalize from the key
                G = int(rng.integers(2, 5))
                cfg["steps"][-1]["G"] = G
                key, ocounts, oEb, omask, best = want(H, p)
                key_t = torch.zeros(1, dtype=torch.int64, device=dev)
                torch.cuda.synchronize()              # (torch fills it on ITS stream; the pair's context may work on another)
                keys, ok = [], True
                for r in range(G):
                    b, cnt = S.shard_range(H, r, G)
                    q = S.default_params(n, num_hypotheses=H, seed=p.seed, kernel=p.kernel, jacobi_sweeps=p.jacobi_sweeps, threshold=p.threshold, hyp_begin=b, hyp_count=cnt)
                    pair.ransac_score(q, key_out=key_t)
                    state["ctx"].synchronize()
                    ok = ok and np.array_equal(pair.get_inlier_counts(cnt), ocounts[b:b + cnt]) and int(key_t.item()) == pair.get_key()
                    keys.append(pair.get_key())
                ok = ok and max(keys) == key
                key_t[0] = max(keys)
                torch.cuda.synchronize()
                pair.ransac_finalize_key(q, key_t)
                ok = ok and pair.get_best() == best and same_bits(pair.get_E(), oEb) and np.array_equal(pair.get_inlier_mask(), omask)
            else:                                     # caller-supplied candidates: hypotheses of another sampler seed, a few of them scaled / negated / transposed
                _, _, oE = O.ransac_range(state["X0"], state["X1"], 0, H, p.threshold, 0, seed=p.seed ^ 0x5A5A, want_E=True)
                Es = oE.reshape(H, 9).copy()
                k = rng.integers(0, H, max(1, H // 16))
                Es[k] *= np.float32(rng.choice([-1.0, 0.25, 3.0]))
                k = rng.integers(0, H, max(1, H // 16))
                Es[k] = Es[k].reshape(-1, 3, 3).transpose(0, 2, 1).reshape(-1, 9)
                if p.kernel == S.KERNEL_AUTO:
                    p.kernel = S.KERNEL_PREFILTER
                pair.ransac_score_candidates(p, to_dev(torch, dev, Es.reshape(-1)))
                with np.errstate(invalid="ignore", over="ignore"):
                    oc = np.array([O.count_inliers_fast(Es[h].reshape(3, 3), state["X0"], state["X1"], np.float32(p.threshold)) for h in range(H)], np.int32)
                bi = int(np.argmax(oc))
                ok = np.array_equal(pair.get_inlier_counts(H), oc) and pair.get_key() == O.pack_key(int(oc[bi]), bi)
            state["have_E"] = op != "candidates" and state.get("have_E", False) or op in ("estimate", "pipelined", "shards")
            if not ok:
                cfg["failed_step"] = step
                cfg["launch"] = pair.last_launch()
                pair.close()
                return False, cfg
        pair.close()
        if state["ctx"] is not ctx:
            state["ctx"].close()
        return True, cfg

    def pose_round():
        """fillXU -> estimateE (few hypotheses) -> poses + triangulation, one launch (sfm_pose_chain) or the three calls, both pose
        modes: candidates, inverses, index and every triangulated point against the oracle, bit for bit.  Flavours: plain /
        noise-free / duplicates / pure rotation-like (tiny baseline) / a NaN pixel (candidates and points may then be NaN: compared
        as bits all the same)."""
        n = int(rng.choice([rng.integers(8, 120), rng.integers(120, 2500)]))
        H = int(rng.integers(1, 96))
        mode = int(rng.choice([S.POSE_REFERENCE, S.POSE_CORRECT]))
        chain = bool(rng.random() < 0.5)
        flavour = str(rng.choice(["plain", "plain", "clean", "dup", "nan"]))
        sseed = int(rng.integers(1, 1 << 30))
        scene = synth.two_view_scene(n, seed=sseed, noise_px=0.0 if flavour == "clean" else float(rng.choice([0.3, 2.0])),
                                     outlier_frac=0.0 if flavour == "clean" else float(rng.choice([0.0, 0.4])))
        sift = scene["sift"].copy()
        if flavour == "dup":
            k = int(rng.integers(1, max(2, n // 2)))
            src = rng.integers(0, n, k); dst = rng.integers(0, n, k)
            sift[dst] = sift[src]
        if flavour == "nan":
            sift["match_xpos"][rng.integers(0, n, max(1, n // 50))] = np.nan
        cfg = dict(path="pose", n=n, H=H, mode=mode, chain=chain, flavour=flavour, seed=sseed)
        pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
        pair.fillXU(to_dev(torch, dev, sift))
        pair.estimateE(S.default_params(n, num_hypotheses=H, seed=sseed & 0xFFFF))
        if chain:
            pair.pose_chain(mode)
        else:
            pair.computePosecandidates(mode); pair.choosePose(mode); pair.linear_triangulation(mode)
        _, _, X0, X1 = O.fill_xu(sift, scene["Kinv"])
        oP = O.pose_candidates(pair.get_E(), mode)
        oind, oPinv, _, _ = O.choose_pose(X0, X1, oP, mode, sweeps=8)
        try:
            ind = pair.get_pose_index()
        except S.SfmError as e:                       # a singular chosen candidate is reported, with the index the oracle has too
            if e.code != S.E_SINGULAR: raise
            return True, cfg
        ok = ind == oind and same_bits(pair.get_pose_candidates(), oP) and same_bits(pair.get_pose_inverses(), oPinv)
        if ok:
            ok = same_bits(pair.get_points(), O.triangulate(X0, X1, oPinv[oind] if mode == S.POSE_REFERENCE else oP[oind], sweeps=8))
        return ok, cfg

    def match_round():
        n1 = int(rng.choice([rng.integers(1, 100), rng.integers(100, 3000)])); n2 = int(rng.choice([rng.integers(1, 100), rng.integers(100, 3000)]))
        sseed = int(rng.integers(1, 1 << 30))
        d = synth.descriptors(max(n1, n2), seed=sseed, noise=float(rng.choice([0.0, 0.05, 0.3])), sparsity=float(rng.choice([0.0, 0.7])))
        d1 = np.ascontiguousarray(d[0][:n1]); d2 = np.ascontiguousarray(d[1][:n2]).copy()
        if rng.random() < 0.3 and n2 > 4:             # exact duplicates in the database: tie rule
            d2[rng.integers(0, n2, n2 // 4)] = d2[rng.integers(0, n2, n2 // 4)]
        t1, t2 = to_dev(torch, dev, d1), to_dev(torch, dev, d2)
        best = torch.empty(n1, dtype=torch.float32, device=dev); sec = torch.empty(n1, dtype=torch.float32, device=dev)
        idx = torch.empty(n1, dtype=torch.int32, device=dev)
        ctx.match_soa(t1, n1, 128, t2, n2, 128, best, sec, idx)
        torch.cuda.synchronize()
        ob, os_, oi = O.match_desc(d1, d2)
        ok = np.array_equal(idx.cpu().numpy(), oi) and same_bits(best.cpu().numpy(), ob) and same_bits(sec.cpu().numpy(), os_)
        return ok, dict(path="match", n1=n1, n2=n2, seed=sseed)

    def homography_round():
        n = int(rng.integers(8, 4000)); L = int(rng.choice([rng.integers(1, 100), rng.integers(100, 4000)]))
        sseed = int(rng.integers(1, 1 << 30))
        sc = synth.homography_scene(n, seed=sseed, noise_px=float(rng.choice([0.0, 0.7, 3.0])), outlier_frac=float(rng.choice([0.0, 0.35, 0.9])))
        thr = float(rng.choice([1.0, 5.0, 25.0])); hs = int(rng.integers(0, 1 << 30))
        ms, ma = float(rng.choice([0.0, 0.85])), float(rng.choice([1.0, 0.95]))
        H, nm = ctx.find_homography(to_dev(torch, dev, sc["sift"]), n, num_loops=L, min_score=ms, max_ambiguity=ma, thresh=thr, seed=hs)
        oH, onm = O.find_homography(sc["sift"], L, ms, ma, thr, hs)
        return (nm == onm and same_bits(H, oH)), dict(path="homography", n=n, loops=L, thr=thr, seed=sseed, hseed=hs, min_score=ms, max_ambiguity=ma)

    def sift_round():
        w = int(rng.integers(24, 700)); h = int(rng.integers(24, 500))
        octaves = int(rng.integers(1, 6)); up = bool(rng.random() < 0.2)
        thresh = float(rng.choice([1.0, 2.0, 3.5, 6.0])); blur = float(rng.choice([0.0, 0.5, 1.0])); lowest = float(rng.choice([0.0, 1.5]))
        sseed = int(rng.integers(1, 1 << 30)); max_pts = int(rng.choice([64, 1000, 8192]))
        img = synth.image(w, h, seed=sseed, blobs=int(rng.integers(5, 300)))
        if rng.random() < 0.2:
            img = img + 20.0 * synth.normal(sseed, w * h).reshape(h, w).astype(np.float32)
        img = np.ascontiguousarray(img, np.float32)
        pitch = (w + 127) // 128 * 128
        pad = np.zeros((h, pitch), np.float32); pad[:, :w] = img
        d_sift = torch.zeros((max_pts, 576), dtype=torch.uint8, device=dev)
        npts, nstored = ctx.extract_sift(d_sift, max_pts, torch.from_numpy(pad).to(dev), w, h, pitch, num_octaves=octaves, init_blur=blur,
                                         thresh=thresh, lowest_scale=lowest, scale_up=up)
        pts = d_sift.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)
        opts, onp, ons = O.extract_sift(img, octaves, blur, thresh, lowest, up, max_pts)
        ok = (npts, nstored) == (onp, ons)
        if ok:
            a = pts[:nstored]; b = opts[:nstored]
            ok = all(same_bits(a[f], b[f]) for f in ("xpos", "ypos", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data"))
        return ok, dict(path="sift", w=w, h=h, octaves=octaves, up=up, thresh=thresh, blur=blur, lowest=lowest, seed=sseed, max_pts=max_pts, npts=int(npts))

    return [("ransac", ransac_round, 0.37), ("calls", calls_round, 0.12), ("pose", pose_round, 0.14), ("match", match_round, 0.13), ("homography", homography_round, 0.12),
            ("sift", sift_round, 0.12)]


def run_rounds(table, rng, budget_s=None, max_rounds=None, max_bad=20):
    rounds = {name: 0 for name, _, _ in table}
    bad = []
    t0 = time.time()
    done = 0
    while (budget_s is None or time.time() - t0 < budget_s) and (max_rounds is None or done < max_rounds):
        r = rng.random(); acc = 0.0
        for name, fn, pr in table:
            acc += pr
            if r < acc:
                break
        try:
            ok, cfg = fn()
        except Exception as e:                        # an exception in a legal configuration is a finding too
            ok, cfg = False, dict(path=name, exception=repr(e))
        rounds[name] += 1
        done += 1
        if not ok:
            bad.append(cfg)
            if len(bad) >= max_bad:
                break
    return rounds, bad, time.time() - t0


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    budget = float(args[0]) if len(args) > 0 else 60.0
    seed = int(args[1]) if len(args) > 1 else 1
    lib = "product"
    if "--lib" in sys.argv:
        lib = sys.argv[sys.argv.index("--lib") + 1]
    import torch
    if lib == "ab":
        import cuda_sfm_amd_ab as S
    else:
        import cuda_sfm_amd as S
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(seed)
    rounds, bad, secs = run_rounds(make_rounds(S, torch, dev, ctx, rng), rng, budget_s=budget)
    print(json.dumps({"seconds": round(secs, 1), "seed": seed, "lib": lib, "rounds": rounds, "mismatches": bad}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
