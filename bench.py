#!/usr/bin/env python3
"""bench.py -- RANSAC E-matrix hypotheses/sec on MI355X (BASELINE.json metric).

One "step" = one estimateE over a synthetic two-view scene (default: 4096 matches, 2^20 hypotheses -- the "4k matches"
configuration of the metric) with the hypothesis ids sharded over the N ranks (strong scaling), one 8-byte
all-reduce(max) selecting the winner and every rank finalizing E + inlier mask.  Inputs are resident in HBM before the
timed region.  Before the W warm-up steps the script runs enough untimed steps to have done 20 in total, so that short
invocations (--warmup 3) do not time the clock ramp; the timed region is exactly K steps.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config headline|c3|c4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no RANK in the environment launches the N ranks itself: the parent process
(which never imports torch and never touches HIP) starts one child per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
set, relays rank 0's JSON line and exits non-zero as soon as any child fails.

What the line carries besides the contract's keys:
  result.parity_vs_oracle   N = 1: EVERY inlier count of the step's H hypotheses, the arg-max key (first maximum), the winner's
                            E (bit for bit) and the inlier mask against the CPU oracle (oracle/ is the checker, never the path).
                            N > 1: every rank must report nccl_ranks == N and the same (key, E, mask) -- else exit code 1.
  roofline                  for the kernel that ran.  The matrix-core pre-filter kernel is priced against its own issue floor
                            (frac <= 1 by construction); the SURVEY 8(d) figure (38 FLOP per pair) stays as
                            `algorithmic_equiv_tflops`.  Counter figures are QUOTED from profiles/<round>_traffic.json and only
                            when the hash of the kernel sources recorded there equals the hash of the sources in this tree.
  cpu_baseline              the oracle's vectorised port on the host cores, bounded sample (its first H hypotheses double as
                            the parity check above).
  extra                     N = 1 only, short runs of the other BASELINE configurations so that each has a clock from the same
                            invocation: c3, c4 (one GPU), the share of one of eight ranks, the descriptor matcher at 2048^2 and
                            16384^2, the dino pair end to end (configs[1]) and the 36-view dino ring / all 630 pairs
                            (configs[4]), each with its own parity boolean.  --no-extra skips them.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_MATCHES = 4096              # "4k matches" of the BASELINE metric
TOTAL_HYPS = 1 << 20          # hypotheses per step over the whole job (BASELINE configs[3] count)
FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 256 FLOP/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FP16_MFMA_PEAK_TFLOPS = 2516.6  # dense fp16 MFMA: 256 CU x 4096 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md, no sparsity)
PEAK_CLOCK_MHZ = 2400.0
NUM_SIMDS = 1024              # 256 CU x 4
FLOP_PER_POINT = 38           # residual of one (hypothesis, point)
FLOP_PER_HYP = 720            # A^T A normal equations
ALG_BYTES_PER_HYP = 72.0       # 32 B indices + 36 B E + 4 B count
# ransac_score_prefilter, per 1024 (hypothesis, point) pairs, what cannot be removed from the formulation a launch runs:
# the fp16 MFMAs of the contraction and the scan that turns every accumulator into one reject bit -- "pack" (round 6, the product; with per-tile
# or, reserved[3] == 6, per-hypothesis band constants), "band" (round 5, reserved[3] == 5), "G" (rounds 2-4, reserved[3] == 4)
# scoring-kernel rules: (vector-issue cycles of a SIMD per 1024 pairs for the irreducible scan, fp16 MFMAs per 1024 pairs)
#   pack: half a v_cvt_scalef32_2xpk16_bf6_f32 (64.6 cycles per 2048 pairs); band: 16 v_alignbit_b32 at 4.24; G: 16 v_fma_f32 at 2.54 + 16 v_alignbit_b32
PF_RULES = {"pack": (32.3, 2), "band": (67.8, 2), "G": (108.5, 3)}
PF_MFMA_CYCLES = 32           # issue interval of one 32x32x16 f16 MFMA on a SIMD (8 passes x 4 cycles; profiles/r02_mfma_rate_probe.txt)
PF_SOURCES = ("ransac_prefilter.hip", "prefilter_math.hpp", "prefilter_record.hpp", "ransac.hip", "ransac_device.hpp", "device_math.hpp")
TRAFFIC_JSON = os.path.join("profiles", "r06_traffic.json")
PUBLISHED_ESTIMATE_E_MS = 24.12     # img/data.xlsx B5 / README.md:54 of the reference: estimateE on the dino pair, GTX 1080 Ti

# BASELINE.json configs that are RANSAC workloads (configs[1] and [4] are pipelines: see `extra`)
CONFIGS = {
    "headline": (N_MATCHES, TOTAL_HYPS, "the metric's '4k matches' configuration"),
    "c3": (16384, 65536, "BASELINE configs[2]: synthetic 16k-match pair, 65k hypotheses"),
    "c4": (16384, 1 << 20, "BASELINE configs[3]: synthetic 16k matches, 1M hypotheses (sharded over --gpus)"),
    "c5": (0, 0, "BASELINE configs[4]: 36-view dino ring, all pairwise estimateE + triangulation, view pairs streamed over --gpus"),
}
KERNEL_NAMES = {1: "ransac_score_waves", 2: "ransac_fused_waves", 3: "ransac_score_mfma", 4: "ransac_score_prefilter"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # 100 x 0.6 ms: long enough for the clocks to settle (20 steps read 4 % slower)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="headline")
    ap.add_argument("--matches", type=int, default=None, help="overrides the preset's match count")
    ap.add_argument("--hyps", type=int, default=None, help="overrides the preset's hypothesis count")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--sweeps", type=int, default=-1, help="null-vector solver: -1 library default, 0 Householder, k > 0 Jacobi sweeps")
    ap.add_argument("--comm", choices=["rccl", "rccl-serial", "torch"], default="rccl",
                    help="multi-GPU exchange step (N > 1): rccl = sfm_estimate_E_sharded_pipelined (include/sfm_amd_comm.h: "
                         "ncclAllReduce(max, u64) + finalize on the communicator's exchange stream, overlapped with the next "
                         "step's scoring); rccl-serial = sfm_estimate_E_sharded (everything on ONE stream, no overlap); "
                         "torch = torch.distributed.all_reduce")
    ap.add_argument("--serial", action="store_true",
                    help="one estimateE at a time (sfm_estimate_E / sfm_estimate_E_sharded) instead of the two-slot pipelined calls "
                         "(sfm_estimate_E_pipelined / sfm_estimate_E_sharded_pipelined) in which consecutive steps overlap on the device")
    ap.add_argument("--timed-events", choices=["auto", "on", "off"], default="auto",
                    help="HIP events around every kernel INSIDE the timed region (auto: only with --serial; pipelined steps are sampled by serial launches right after it)")
    ap.add_argument("--reserved", type=int, nargs="*", default=[], help="sfm_ransac_params.reserved[] A/B switches (profiles/): runs on libsfm_amd_ab.so, never part of a judged line")
    ap.add_argument("--pairs", choices=["all", "ring"], default="all", help="--config c5: all 630 unordered pairs of the 36 views, or the 36 ring pairs")
    ap.add_argument("--regions", type=int, default=5, help="timed regions of K steps each: the contractual one + (regions - 1) more, reported as ms_per_step_regions")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg (and with it the full-oracle parity check)")
    ap.add_argument("--no-variants", action="store_true", help="skip the short run with the other null-vector solver")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other BASELINE configurations")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget of the cpu_baseline sample")
    ap.add_argument("--exchange-probe", action="store_true", help="internal: the child process of a one-GPU run that times the exchange step alone")
    ap.add_argument("--no-exchange-probe", action="store_true", help="one GPU: do not start the child process that times the exchange step through a one-rank RCCL communicator "
                                                                     "(skipped by itself under rocprofv3 / any preloaded tool)")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="--gpus N started without a launcher: seconds after which bench.py's own launcher reports every rank's last stage, stops the ranks and exits 124")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# stage lines: every rank says where it is (stderr, and the file the launcher reads when it has to give up on a rank)
# ------------------------------------------------------------------------------------------------------------------
_T0 = time.time()


def stage(msg):
    r, w = os.environ.get("RANK", "0"), os.environ.get("WORLD_SIZE", "1")
    line = f"[bench rank {r}/{w} +{time.time() - _T0:7.2f}s] stage {msg}"
    if int(w) > 1 or os.environ.get("SFM_BENCH_STAGES"):
        print(line, file=sys.stderr, flush=True)
    path = os.environ.get("SFM_BENCH_STAGE_FILE")
    if path:
        try:
            with open(path, "a") as f:
                f.write(line + "\n")
        except OSError:
            pass


# ------------------------------------------------------------------------------------------------------------------
# launcher: parent of the N ranks.  Nothing in here imports torch or the HIP library.
# ------------------------------------------------------------------------------------------------------------------
def visible_gpus():
    """Device count, asked of a throw-away child process so that the launcher itself never initialises the GPU runtime."""
    code = "import torch; print(torch.cuda.device_count())"
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except (subprocess.SubprocessError, ValueError, IndexError):
        return 0


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, env_extra=None, command=None, timeout=900.0):
    """Starts n child processes of this script (or of `command`), one per rank, and relays rank 0's stdout.
    Returns the exit code for the parent: 0 only if every rank exited 0.  The children are polled: the first rank that
    exits non-zero ends the job at once (the others, possibly blocked in a collective that can no longer complete, are
    terminated) instead of leaving the launcher waiting for the time-out.  On the time-out (default 900 s: inside the
    1800 s the driver gives a SCALE step, so that a hang in ncclCommInitRank or the first collective is REPORTED, not cut off)
    the launcher prints the last stage line of every rank (their SFM_BENCH_STAGE_FILE), stops exactly the processes it started
    and returns 124."""
    import tempfile
    port = free_port()
    procs = []
    stage_dir = tempfile.mkdtemp(prefix="bench_stages_")
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "SFM_BENCH_STAGE_FILE": os.path.join(stage_dir, f"rank{r}.stage")})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this host driver
        if env_extra:
            env.update(env_extra)
        cmd = command if command is not None else [sys.executable, os.path.abspath(__file__)] + list(argv)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0 = []
    drain = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)   # keeps rank 0's pipe from filling up
    drain.start()
    deadline = time.time() + timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if failed:
            r, c = failed[0]
            print(f"bench.py: rank {r} exited with code {c}; stopping the other ranks", file=sys.stderr)
            rc = c if c > 0 else 1
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            print(f"bench.py: time-out after {timeout:.0f} s; last stage of every rank:", file=sys.stderr)
            for r in range(n):
                last = "(no stage line: the rank never got as far as importing the libraries)"
                try:
                    with open(os.path.join(stage_dir, f"rank{r}.stage")) as f:
                        lines = f.read().splitlines()
                    if lines:
                        last = lines[-1]
                except OSError:
                    pass
                state = "running" if codes[r] is None else f"exited {codes[r]}"
                print(f"bench.py:   rank {r} ({state}): {last}", file=sys.stderr)
            rc = 124
            break
        time.sleep(0.2)
    for r, p in enumerate(procs):
        if p.poll() is None:                      # exactly the PIDs started above
            p.terminate()
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
            rc = rc or 125
    drain.join(timeout=5)
    for line in out0:                             # library banners (gloo / RCCL print to stdout) go to stderr: stdout carries the JSON line only
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line if line.endswith("\n") else line + "\n")
    sys.stdout.flush()
    try:
        for r in range(n):
            os.remove(os.path.join(stage_dir, f"rank{r}.stage"))
    except OSError:
        pass
    try:
        os.rmdir(stage_dir)
    except OSError:
        pass
    return rc


def launcher_main(args, argv):
    have = visible_gpus()
    if have < args.gpus and os.environ.get("SFM_BENCH_DIST_BACKEND", "nccl") == "nccl":      # (tests fold the ranks onto the GPUs there are)
        print(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s); nothing was run", file=sys.stderr)
        return 2
    return launch_ranks(args.gpus, argv, timeout=args.launch_timeout)


# ------------------------------------------------------------------------------------------------------------------
# CPU oracle (rank 0, N = 1 only): the baseline leg and the parity check share one sweep
# ------------------------------------------------------------------------------------------------------------------
def load_oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    return O


def cpu_baseline(O, X0, X1, params, n_matches, H, seconds, scene=None):
    """CPU port of the same algorithm on a bounded sample (oracle/: OpenMP over hypotheses; the scoring loop is the
    vectorised division-free filter + exact fallback).  The sample STARTS with hypotheses 0..H-1, counts kept: that sweep
    is also what the GPU's counts and arg-max key are compared with.  Returns (baseline dict, oracle key, oracle counts)."""
    avail = len(os.sched_getaffinity(0))
    run = O.ransac_range_fast

    def rate_of(threads, hyps):
        t0 = time.perf_counter()
        run(X0, X1, 0, hyps, params.threshold, params.jacobi_sweeps, seed=params.seed, want_counts=False, nthreads=threads)
        return hyps / (time.perf_counter() - t0)

    # the affinity mask may promise more CPUs than the container's quota delivers (then more threads are SLOWER): take the
    # thread count that measures best on a short probe and report that number as `cores`
    rate_of(avail, 64 * avail)                            # thread start-up
    cores, rate = avail, 0.0
    for threads in sorted({avail, max(1, avail // 2), max(1, avail // 4), min(avail, 16), 1}, reverse=True):
        r = rate_of(threads, max(256, int(2000 * threads)))
        if r > rate:
            cores, rate = threads, r
    t0 = time.perf_counter()
    key, counts, _ = run(X0, X1, 0, H, params.threshold, params.jacobi_sweeps, seed=params.seed, want_counts=True, nthreads=cores)
    dt = time.perf_counter() - t0
    sample = H
    more = int(min(256 * TOTAL_HYPS, H / dt * max(0.0, seconds - dt)))      # ids beyond H are further hypotheses of the same scene
    if more >= 4096:
        t0 = time.perf_counter()
        run(X0, X1, H, more, params.threshold, params.jacobi_sweeps, seed=params.seed, want_counts=False, nthreads=cores)
        dt += time.perf_counter() - t0
        sample += more
    out = {"value": sample / dt, "unit": "hypotheses/s", "cores": cores, "cpus_in_affinity_mask": avail, "kind": "port",
           "impl": ("oracle/sfm_oracle_fast.c: compiler-vectorised (AVX-512 / AVX2 by CPU) division-free filter + exact fallback, "
                    "count-exact vs the scalar restatement (tests/test_oracle_fast.py)"),
           "sample": f"hypotheses 0..{sample - 1} of the same {n_matches}-match scene ({H} of them with counts kept for the parity check), "
                     f"{dt:.1f} s, OpenMP x{cores}",
           # north_star also names OpenCV's cv::findEssentialMat (a 5-point solver: wall-time sanity, not a parity target)
           "opencv_findEssentialMat": opencv_leg(scene)}
    return out, key, counts


def opencv_leg(scene):
    """cv::findEssentialMat(pts1, pts2, K, RANSAC, 0.999, 1.0) on the step's matches and cv::BFMatcher on a 2048 x 2048 descriptor
    set, timed on the host cores whenever cv2 imports (SURVEY 8(d) iii: a 5-point solver with adaptive termination -- a wall-time
    and pose sanity figure, not a parity target and not the cpu_baseline value).  This image has no OpenCV: the leg then says so."""
    try:
        import cv2
    except ImportError:
        return "unavailable: OpenCV (cv2) is not installed in this image"
    import numpy as np
    try:
        sift = scene["sift"]
        p1 = np.ascontiguousarray(np.stack([sift["xpos"], sift["ypos"]], 1), np.float64)
        p2 = np.ascontiguousarray(np.stack([sift["match_xpos"], sift["match_ypos"]], 1), np.float64)
        K = np.ascontiguousarray(scene["K"], np.float64).reshape(3, 3)
        cv2.setNumThreads(len(os.sched_getaffinity(0)))
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            E, mask = cv2.findEssentialMat(p1, p2, K, method=cv2.RANSAC, prob=0.999, threshold=1.0)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out = {"findEssentialMat_ms": 1e3 * best, "matches": int(len(p1)), "inliers": int(mask.sum()) if mask is not None else None,
               "args": "cv2.RANSAC, prob 0.999, threshold 1.0 px (5-point solver, adaptive iteration count)",
               "threads": cv2.getNumThreads(), "version": cv2.__version__}
        sys.path.insert(0, ROOT)
        from cuda_sfm_amd_synth import synth
        d1, d2, _ = synth.descriptors(2048)
        bf = cv2.BFMatcher(cv2.NORM_L2)
        t0 = time.perf_counter()
        bf.knnMatch(d2, d1, k=2)
        out["BFMatcher_knn2_2048x2048_ms"] = 1e3 * (time.perf_counter() - t0)
        return out
    except Exception as e:                            # noqa: BLE001 -- a reported side figure must not take the line down
        return f"cv2 {getattr(cv2, '__version__', '?')} imported but the leg failed: {type(e).__name__}: {e}"


def regions_summary(region_s, steps):
    ms = sorted(1e3 * r / steps for r in region_s)
    med = ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2])
    return {"regions": len(ms), "steps_per_region": steps, "median": med, "min": ms[0], "max": ms[-1],
            "all": [round(1e3 * r / steps, 5) for r in region_s],
            "note": "all[0] is the contractual region (= ms_per_step / value); the others follow it back to back, timed the same way"}


def full_parity(O, X0, X1, params, n, okey, ocounts, gpu):
    """gpu = (counts[H], key, E 3x3, mask) of ONE estimateE; okey / ocounts = the oracle's sweep of the same H hypotheses.
    True iff every count, the key (arg-max, first maximum), the winner's E bit for bit and the mask agree."""
    import numpy as np
    counts, key, E, mask = gpu
    ocnt, ohyp = O.unpack_key(okey)
    if not (np.array_equal(counts, ocounts) and int(key) == int(okey)):
        return False
    if ocnt != int(ocounts.max()) or ohyp != int(np.argmax(ocounts)):       # the oracle's own key must be the first maximum
        return False
    oE = O.hypothesis_E(X0, X1, O.sample8(params.seed, ohyp, n), params.jacobi_sweeps)
    c2, omask = O.count_inliers(oE, X0, X1, params.threshold)                # scalar restatement for the winner
    return bool(c2 == ocnt and np.array_equal(omask, mask) and np.array_equal(oE.reshape(-1).view(np.uint32), np.ascontiguousarray(E, np.float32).reshape(-1).view(np.uint32)))


# ------------------------------------------------------------------------------------------------------------------
# roofline of the scoring kernel
# ------------------------------------------------------------------------------------------------------------------
def source_hash():
    h = hashlib.sha256()
    for name in PF_SOURCES:
        with open(os.path.join(ROOT, "cuda-sfm_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def quoted_counters(kname, n, local_hyps):
    """Counter figures of the committed rocprofv3 --pmc passes -- quoted, not collected by this run, and only if they were
    collected on the kernel sources of this tree (hash of PF_SOURCES recorded by profiles/make_traffic_json.py)."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_JSON)) as f:
            doc = json.load(f)
        t = doc.get(kname)
        if not t or t["matches"] != n or t["hypotheses"] != local_hyps:
            return None, f"{TRAFFIC_JSON} has no entry for {kname} at {n} x {local_hyps}"
        have = source_hash()
        if doc.get("code_sha256_16") != have:
            return None, (f"{TRAFFIC_JSON} was collected on kernel sources {doc.get('code_sha256_16')}, this tree has {have}: "
                          "not quoted (re-run profiles/collect_r06.sh)")
        t = dict(t)
        solve = doc.get("ransac_solve_lanes1_qr") or doc.get("ransac_solve_lanes2") or {}
        if solve.get("fetch_kb") is not None and solve.get("write_kb") is not None:
            t["solve_fetch_kb"], t["solve_write_kb"] = solve["fetch_kb"], solve["write_kb"]
        t["source"] = f"{TRAFFIC_JSON} (rocprofv3 --pmc passes of this command on sources {have}; quoted, not collected by this run)"
        return t, None
    except (OSError, KeyError, ValueError) as e:
        return None, f"{TRAFFIC_JSON}: {type(e).__name__}"


def reserved_rule(reserved):
    """Which scoring rule the lab-bench switches select (ransac_prefilter.hip: prefilter_rule)."""
    r = list(reserved) + [0] * 4
    if r[3] == 4 or r[3] >= 16:
        return "G"
    if r[3] == 5 or r[1] in (5, 7, 9):
        return "band"
    return "pack"


def roofline_block(kernel_id, n, local_hyps, score_s, solve_s, clock_mhz, measured_in, rule="pack"):
    kname = KERNEL_NAMES.get(kernel_id, "?")
    pairs = float(local_hyps) * n
    alg_flops = pairs * FLOP_PER_POINT
    alg_tflops = alg_flops / score_s / 1e12 if score_s > 0 else 0.0
    quoted, why_not = quoted_counters(kname, n, local_hyps)
    common = {"kernel": kname, "avg_launch_ms": 1e3 * score_s, "solve_kernel_avg_ms": 1e3 * solve_s, "measured_in": measured_in,
              "shader_clock_mhz": clock_mhz, "pairs_per_launch": pairs,
              "algorithmic_equiv_tflops": alg_tflops,
              "algorithmic_equiv_note": "SURVEY 8(d): 38 FLOP per (hypothesis, point) pair over the launch time; the FP32 peak is 157.3 TFLOP/s",
              "pipeline_algorithmic_equiv_tflops": (float(local_hyps) * (FLOP_PER_HYP + FLOP_PER_POINT * n)) / max(score_s + solve_s, 1e-12) / 1e12}
    if kernel_id == 4:
        # ONE floor: the work the kernel executes per 1024 pairs on its binding unit.  All three rules issue their fp16 MFMAs (32 cycles of a
        # SIMD's matrix pipe each) next to the scan's vector instructions; the floor is whichever of the two takes longer at 2.4 GHz:
        #   pack  (round 6, the product): 2 MFMAs = 64 cycles against 1/2 v_cvt_scalef32_2xpk16_bf6_f32 = 32.3 cycles  -> MFMA-bound
        #   band  (round 5, lab bench):   2 MFMAs = 64 cycles against 16 v_alignbit_b32 = 67.8 cycles                  -> scan-bound (by 6 %)
        #   G     (rounds 2-4, lab bench): 3 MFMAs = 96 cycles against 16 v_fma + 16 v_alignbit = 108 cycles
        # (instruction costs measured on the MI355X: profiles/r05_valu_rate_table.txt, r06_cvt_pack_probe.txt)
        clock = PEAK_CLOCK_MHZ * 1e6
        scan_cycles, mfmas = PF_RULES[rule]
        valu_floor = pairs / 1024.0 * scan_cycles / NUM_SIMDS / clock
        mfma_floor = pairs / 1024.0 * mfmas * PF_MFMA_CYCLES / NUM_SIMDS / clock
        floor = max(valu_floor, mfma_floor)
        mfma_tflops = pairs / 1024.0 * mfmas * 32768.0 / score_s / 1e12 if score_s > 0 else 0.0
        out = {"bound": "mfma" if mfma_floor >= valu_floor else "valu_issue",
               "bound_detail": (f"{rule} rule: {mfmas} x v_mfma_f32_32x32x16_f16 (32768 FLOP, 32 cycles of a SIMD's matrix pipe each) per 1024 (hypothesis, point) pairs; "
                                f"the scan's vector instructions cost {scan_cycles:.1f} cycles of the SIMD's vector issue per 1024 pairs "
                                "(profiles/r06_cvt_pack_probe.txt, r05_valu_rate_table.txt); floor_ms = the larger of the two at 2.4 GHz; "
                                "achieved / peak are the fp16 MFMA FLOP the kernel executes against the dense fp16 MFMA peak (no sparsity)"),
               "achieved": mfma_tflops, "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": (floor / score_s) if score_s > 0 else 0.0,
               "floor_ms": 1e3 * floor, "mfma_floor_ms": 1e3 * mfma_floor, "valu_scan_floor_ms": 1e3 * valu_floor,
               "frac_at_sustained_clock": (floor / score_s * PEAK_CLOCK_MHZ / clock_mhz) if (score_s > 0 and clock_mhz > 0) else None,
               "rule": rule,
               # SURVEY 8(d)'s figure, kept for the record and flagged: it prices 38 FLOP per pair that this kernel does not execute
               "survey_8d_ratio": alg_tflops / FP32_PEAK_TFLOPS,
               "survey_8d_note": ("> 1: not a roofline fraction -- 38 FLOP x pairs / launch time over the 157.3 TFLOP/s FP32 peak; the kernel is a work-reducing "
                                  "pre-filter (fp16 MFMA contraction + one-bit test for every pair, the exact 38-FLOP test for the ~1.3 % that survive)")}
    else:
        out = {"bound": "valu_fp32",
               "bound_detail": ("FP32 vector ALU issue (v_pk_fma_f32 and friends); 157.3 TFLOP/s = 256 CU x 256 FLOP/clk x 2.4 GHz, "
                                "numerically the dense f32 MFMA peak the bench contract prices compute against"),
               "achieved": alg_tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": alg_tflops / FP32_PEAK_TFLOPS,
               "frac_at_sustained_clock": (alg_tflops / (FP32_PEAK_TFLOPS * clock_mhz / PEAK_CLOCK_MHZ)) if clock_mhz > 0 else None}
    out.update(common)
    if quoted:
        out["traffic"] = 1024.0 * (quoted["fetch_kb"] + quoted["write_kb"])
        out["traffic_source"] = quoted["source"]
        # 32 B of sample indices in (here: derived from the id, never read) + 36 B E + 4 B count out per hypothesis,
        # + the 16 N-byte point set once.  Per-tile re-reads of E / the records are implementation, not algorithm.
        out["traffic_algorithmic_bytes"] = ALG_BYTES_PER_HYP * float(local_hyps) + 16.0 * n
        out["traffic_step_bytes"] = 1024.0 * (quoted["fetch_kb"] + quoted["write_kb"] + quoted.get("solve_fetch_kb", 0.0) + quoted.get("solve_write_kb", 0.0))
        out["traffic_step_over_algorithmic"] = out["traffic_step_bytes"] / out["traffic_algorithmic_bytes"]
        out["traffic_note"] = ("traffic = FETCH_SIZE + WRITE_SIZE of the scoring kernel per launch; traffic_step_bytes adds the lane-solve kernel of the same "
                               "step; algorithmic = 72 B per hypothesis + 16 B per match (SURVEY 8(d))")
        out["issue_profiled"] = {k: quoted[k] for k in ("valu_busy_frac", "mfma_busy_frac", "lds_busy_frac", "valu_insts_per_launch", "lds_bank_conflict_frac", "kernel_cycles",
                                                             "wave_issuing_frac", "wave_parked_at_waitcnt_frac", "wave_issue_stalled_frac") if k in quoted}
    else:
        out["traffic"] = None
        out["traffic_source"] = why_not
    return out


# ------------------------------------------------------------------------------------------------------------------
# extra: the other BASELINE configurations, short runs (rank 0, N = 1)
# ------------------------------------------------------------------------------------------------------------------
def extra_ransac(S, synth, O, ctx, dev, torch, np, name, n, H, steps, hyp_count=None):
    """A short pipelined run of another RANSAC configuration + the full-oracle parity check of one call."""
    scene = synth.two_view_scene(n)
    d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.fillXU(d_sift)
    p = S.default_params(n, num_hypotheses=H)
    if hyp_count is not None:
        p.hyp_begin, p.hyp_count = 0, hyp_count
    local = hyp_count if hyp_count is not None else H
    # untimed wake-up as in the headline run: ~40 ms of this configuration's own work (a short run right after another kernel mix
    # reads 5-10 % slow: the clock has not settled), then the timed steps, then 20 serial launches for the kernel's own duration
    est_step_s = local * n / 7e12 + 20e-6
    for _ in range(min(2000, max(5, int(0.04 / est_step_s)))):
        pair.estimateE_pipelined(p)
    pair.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pair.estimateE_pipelined(p)
    pair.flush(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ks = max(5, min(steps, 50))
    for _ in range(3):
        pair.estimateE(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ks):
        pair.estimateE(p)                                 # one call at a time
    torch.cuda.synchronize()
    dt_serial = (time.perf_counter() - t0) / ks
    for _ in range(3):
        pair.ransac_score(p)
    ctx.synchronize()
    ctx.kernel_timing(True)
    for _ in range(20):
        pair.ransac_score(p)
    solve_ms, score_ms, calls = ctx.kernel_timing_read()
    ctx.kernel_timing(False)
    pair.estimateE(p)
    launch = pair.last_launch()
    got = (pair.get_inlier_counts(local).copy(), pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy())
    hyp, cnt = pair.get_best()
    out = {"matches": n, "hypotheses_per_step": local, "ms_per_step": 1e3 * dt, "serial_ms_per_step": 1e3 * dt_serial, "hypotheses_per_s": local / dt, "steps": steps,
           "kernel": KERNEL_NAMES.get(launch["kernel"], "?"), "score_kernel_ms": score_ms / max(calls, 1), "solve_kernel_ms": solve_ms / max(calls, 1),
           "best_hypothesis": hyp, "inliers": cnt}
    if launch["kernel"] == 4 and score_ms > 0:
        r = roofline_block(4, n, local, score_ms / 1e3 / calls, solve_ms / 1e3 / calls, pair.last_clock_mhz(), "20 serial launches")
        out["roofline_frac"] = r["frac"]
    pair.close()
    if O is not None:
        def check():                                      # run after ALL GPU timings (see run_extras)
            _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
            t0 = time.perf_counter()
            okey, ocounts, _ = O.ransac_range_fast(X0, X1, 0, local, p.threshold, p.jacobi_sweeps, seed=p.seed)
            out["oracle_sweep_s"] = time.perf_counter() - t0
            out["parity_vs_oracle"] = full_parity(O, X0, X1, p, n, okey, ocounts, got)
        out["_check"] = check
    return out


MATCH_TRAFFIC_JSON = os.path.join("profiles", "r06_match_traffic.json")
MATCH_SOURCES = ("match.hip", "match_fused.hip", "match_prefilter.hip", "match_common.hpp", "match_prefilter_math.hpp")


def match_traffic(n, prefixes, ms):
    """HBM bytes per sfm_match_soa call from the committed PMC passes (profiles/pmc_match.sh: FETCH_SIZE / WRITE_SIZE in passes of their
    own, KB), summed over the kernels of the family that ran, and the GB/s that is at the time measured HERE; quoted only when the
    matcher sources are the ones the counters were collected on."""
    try:
        with open(os.path.join(ROOT, MATCH_TRAFFIC_JSON)) as f:
            doc = json.load(f)
        h = hashlib.sha256()
        for name in MATCH_SOURCES:
            with open(os.path.join(ROOT, "cuda-sfm_amd", "csrc", name), "rb") as f:
                h.update(f.read())
        have = h.hexdigest()[:16]
        if doc.get("code_sha256_16") != have:
            return {"note": f"{MATCH_TRAFFIC_JSON} was collected on matcher sources {doc.get('code_sha256_16')}, this tree has {have}: not quoted"}
        ks = {k: v for k, v in doc.get("sizes", {}).get(str(n), {}).items() if any(k.startswith(p) for p in prefixes)}
        if not ks or any(v.get("fetch_kb") is None or v.get("write_kb") is None for v in ks.values()):
            return {"note": f"{MATCH_TRAFFIC_JSON} has no FETCH_SIZE / WRITE_SIZE for this kernel family at n = {n}"}
        fetch = sum(v["fetch_kb"] for v in ks.values()) * 1024.0
        write = sum(v["write_kb"] for v in ks.values()) * 1024.0
        algorithmic = 2.0 * n * 128 * 4 + n * 12.0       # both descriptor sets once + best, second, index
        return {"kernels": sorted(ks), "fetch_bytes": fetch, "write_bytes": write, "algorithmic_bytes": algorithmic,
                "hbm_gb_per_s_at_this_time": (fetch + write) / ms / 1e6, "frac_of_hbm_peak": (fetch + write) / ms / 1e6 / HBM_PEAK_GBS,
                "source": f"{MATCH_TRAFFIC_JSON} (rocprofv3 --pmc passes of profiles/pmc_match.sh on sources {have}; quoted, not collected by this run)"}
    except Exception as e:  # noqa: BLE001 - the counters are optional evidence
        return {"note": f"{MATCH_TRAFFIC_JSON}: {type(e).__name__}"}


def extra_match(S, synth, O, ctx, dev, torch, np, n, reps):
    d1, d2, perm = synth.descriptors(n)
    t1, t2 = torch.from_numpy(d1).to(dev), torch.from_numpy(d2).to(dev)
    best = torch.empty(n, dtype=torch.float32, device=dev); sec = torch.empty_like(best)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.set_match_kernel(S.MATCH_AUTO)
    for _ in range(3):
        ctx.match_soa(t2, n, 128, t1, n, 128, best, sec, idx)
    torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        ctx.match_soa(t2, n, 128, t1, n, 128, best, sec, idx)
    ms = ctx.timer_stop() / reps
    ran = ctx.last_match_kernel()
    res = (best.cpu().numpy().copy(), sec.cpu().numpy().copy(), idx.cpu().numpy().copy())
    flops = 2.0 * n * n * 128
    out = {"n": n, "ms": ms, "kernel": {S.MATCH_EXACT: "exact fp32 MFMA", S.MATCH_PREFILTER: "fp16 MFMA pre-filter + exact candidates (four launches)",
                                    S.MATCH_FUSED: "fp16 MFMA pre-filter + exact candidates (one launch, running threshold)"}.get(ran, str(ran)),
           "algorithmic_tflops": flops / ms / 1e9, "perm_recovered": float((res[2] == perm).mean())}
    if ran == S.MATCH_EXACT:
        out["frac_of_fp32_mfma_peak"] = flops / ms / 1e9 / FP32_PEAK_TFLOPS
    out["traffic"] = match_traffic(n, {S.MATCH_EXACT: ("match_mfma_kernel",), S.MATCH_PREFILTER: ("match_pf_",), S.MATCH_FUSED: ("match_fused",)}.get(ran, ()), ms)
    if O is not None:
        def check():
            # the CPU matcher restates MatchC1 (CudaSift/match.cu:57-71): EVERY query at every size (16384^2: ~6 s of host time),
            # index, best and second score bit for bit
            t0 = time.perf_counter()
            cb, cs, ci = O.match_desc(d2, d1, nthreads=len(os.sched_getaffinity(0)))
            out["oracle_sweep_s"] = time.perf_counter() - t0
            out["parity_vs_oracle"] = bool(np.array_equal(ci, res[2]) and np.array_equal(cb.view(np.uint32), res[0].view(np.uint32))
                                           and np.array_equal(cs.view(np.uint32), res[1].view(np.uint32)))
            out["parity_queries_checked"] = n
        out["_check"] = check
    return out


def dino_pairs_vs_oracle(S, O, np, feats, pairs, res, Kinv):
    """Every pair of `pairs` (view indices into feats = per-view SiftPoint records) through the oracle chain of src/main.cpp:282-307
    against the GPU records res[pid] = [E(9) | chosen pose(16) | pose index, inliers, best hypothesis].  Returns the ids that differ."""
    bad = []
    for pid, (i, j) in enumerate(pairs):
        fi, fj = feats[i], feats[j]
        r = res.get(pid) if hasattr(res, "get") else res[pid]
        if len(fi) < 8:
            if r is not None and r[26] >= 0:
                bad.append(pid)
            continue
        om = O.match_sift(fi.copy(), fj)
        _, _, X0, X1 = O.fill_xu(om, Kinv)
        q = S.default_params(len(fi))
        okey, _, Ec = O.ransac_range(X0, X1, 0, q.num_hypotheses, q.threshold, q.jacobi_sweeps, seed=q.seed, want_E=True)
        ocnt, ohyp = O.unpack_key(okey)
        oP = O.pose_candidates(Ec[ohyp], S.POSE_REFERENCE)
        oind, oPinv, _, _ = O.choose_pose(X0, X1, oP, S.POSE_REFERENCE, 8)
        ok = r is not None and (int(r[26]), int(r[27]), int(r[25])) == (ocnt, ohyp, oind) and \
            np.array_equal(np.ascontiguousarray(r[:9], np.float32).view(np.uint32), Ec[ohyp].reshape(-1).view(np.uint32)) and \
            np.array_equal(np.ascontiguousarray(r[9:25], np.float32).view(np.uint32), np.ascontiguousarray(oPinv[oind], np.float32).reshape(-1).view(np.uint32))
        if not ok:
            bad.append(pid)
    return bad


def extra_dino(S, O, ctx, dev, torch, np):
    """BASELINE configs[1] (dino pair: match + estimateE end to end, 1024 hypotheses) and configs[4] (36-view ring / all 630
    pairs) on the reference's own frames (8-bit grey fixtures under tests/golden/dino)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import read_pnm_grey, dino_frame, DINO_K, DINO_KINV, DINO_SIFT
    if not os.path.exists(dino_frame(35)):
        return {"skipped": "tests/golden/dino fixtures not present"}
    views = [read_pnm_grey(dino_frame(k)) for k in range(36)]
    views8 = [v.astype(np.uint8) for v in views]            # the frames as the 8-bit images they are (sfm_extract_views_u8)
    h, w = views[0].shape
    pitch = (w + 127) // 128 * 128
    out = {}

    def extract(img):
        pad = np.zeros((h, pitch), np.float32); pad[:, :w] = img
        d_sift = torch.zeros((32768, 576), dtype=torch.uint8, device=dev)
        n, _ = ctx.extract_sift(d_sift, 32768, torch.from_numpy(pad).to(dev), w, h, pitch, **DINO_SIFT)
        return d_sift, n

    def timed(fn, reps=50):
        for _ in range(5):
            fn()
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop() / reps

    (s1, n1), (s2, n2) = extract(views[0]), extract(views[1])
    pair = S.ImagePair(ctx, DINO_K, DINO_KINV, 2, n1)
    p = S.default_params(n1, num_hypotheses=1024)               # configs[1]: "~2k matches, 1024 RANSAC hypotheses"
    stage = {"match": timed(lambda: ctx.match(s1, n1, s2, n2)), "fillXU": timed(lambda: pair.fillXU(s1)),
             "estimateE": timed(lambda: pair.estimateE(p)), "pose_chain": timed(lambda: pair.pose_chain())}

    def end_to_end():
        ctx.match(s1, n1, s2, n2); pair.fillXU(s1); pair.estimateE(p); pair.pose_chain()
    stage["match_fillXU_estimateE_pose_chain"] = timed(end_to_end)
    hyp, cnt = pair.get_best()
    c1 = {"features": [n1, n2], "hypotheses": 1024, "ms": stage, "inliers": cnt, "best_hypothesis": hyp,
          "published_estimateE_ms_gtx1080ti": PUBLISHED_ESTIMATE_E_MS, "estimateE_speedup_vs_published": PUBLISHED_ESTIMATE_E_MS / stage["estimateE"],
          "note": "the published 24.12 ms are for H = N/8 = 269 hypotheses; this run does 1024"}
    if O is not None:
        m = s1.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)[:n1]
        f2 = s2.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)[:n2]
        got = (pair.get_inlier_counts(1024).copy(), pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy())

        def check_c1():                                   # deferred: see run_extras
            om = O.match_sift(m.copy(), f2)
            ok = bool(np.array_equal(m["match"], om["match"]) and np.array_equal(m["score"].view(np.uint32), om["score"].view(np.uint32)))
            _, _, X0, X1 = O.fill_xu(om, DINO_KINV)
            okey, ocounts, _ = O.ransac_range_fast(X0, X1, 0, 1024, p.threshold, p.jacobi_sweeps, seed=p.seed)
            c1["parity_vs_oracle"] = bool(ok and full_parity(O, X0, X1, p, n1, okey, ocounts, got))
        c1["_check"] = check_c1
    out["c1_dino_pair"] = c1
    pair.close()

    feats_cache = []

    def view_feats():                                     # the 36 views' features as the GPU extractor delivers them, once
        if not feats_cache:
            for v in views:
                sv, nv = extract(v)
                feats_cache.append(sv.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)[:nv].copy())
        return feats_cache

    # configs[4]: host images in, everything per pair inside the C library (sfm_extract_views + sfm_process_pairs)
    for name, pairs in (("c5_dino_ring_36_pairs", S.ring_pairs(36)), ("c5_dino_all_630_pairs", [(i, j) for i in range(36) for j in range(i + 1, 36)])):
        S.process_views(ctx, views8[:9], DINO_K, DINO_KINV, max_pts=8192, sift=DINO_SIFT, device=dev)
        runs = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res, counts = S.process_views(ctx, views8, DINO_K, DINO_KINV, pairs=pairs, max_pts=8192, sift=DINO_SIFT, device=dev)
            torch.cuda.synchronize(); runs.append(time.perf_counter() - t0)
        e = {"pairs": len(pairs), "done": len(res), "features_per_view": [min(counts), max(counts)], "ms_total": 1e3 * min(runs),
             "ms_per_pair": 1e3 * min(runs) / len(pairs), "ms_runs": [round(1e3 * r, 3) for r in runs],
             "note": "8-bit host images -> device inside the timed region (PCIe-inclusive; float images: +0.5 ms, profiles/r03_ring_bench.txt)"}
        if O is not None:
            def check_c5(e=e, res=res, pairs=pairs):
                # EVERY pair against the oracle chain (MatchC1 restatement -> fillXU -> estimateE -> pose candidates -> choosePose),
                # from the features the GPU extractor delivers for the views: winner, count, E bits, pose index, chosen pose
                t0 = time.perf_counter()
                bad = dino_pairs_vs_oracle(S, O, np, view_feats(), pairs, res, DINO_KINV)
                e["parity_vs_oracle"] = bool(len(res) == len(pairs) and not bad)
                e["parity_pairs_checked"] = len(pairs)
                e["oracle_sweep_s"] = round(time.perf_counter() - t0, 2)
                if bad:
                    e["pairs_that_differ"] = bad[:8]
            e["_check"] = check_c5
        out[name] = e
    return out


def run_extras(S, synth, O, ctx, dev, torch, np, skip):
    """GPU timings of ALL configurations first, the CPU oracle's parity sweeps afterwards: the oracle is OpenMP code whose
    worker threads keep spinning for a while after a parallel region and compete with the thread that launches the next
    configuration (a rank's share is three launches per 0.1 ms step: one descheduling of that thread reads as 0.9 ms per step)."""
    out = {}
    time.sleep(0.5)                                       # (the cpu_baseline leg has just ended: let its workers go to sleep)

    def guarded(name, fn):
        try:
            t0 = time.perf_counter()
            out[name] = fn()
            if isinstance(out[name], dict):
                out[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        except Exception as e:                            # noqa: BLE001 -- an extra must never take the headline line down
            out[name] = {"error": f"{type(e).__name__}: {e}"}

    if "c3" not in skip:
        guarded("c3", lambda: extra_ransac(S, synth, O, ctx, dev, torch, np, "c3", 16384, 65536, 50))
    if "c4" not in skip:
        guarded("c4_one_gpu", lambda: extra_ransac(S, synth, O, ctx, dev, torch, np, "c4", 16384, 1 << 20, 10))
    guarded("headline_share_of_one_of_8_ranks", lambda: extra_ransac(S, synth, O, ctx, dev, torch, np, "rank8", N_MATCHES, TOTAL_HYPS, 100, hyp_count=TOTAL_HYPS // 8))
    guarded("c4_share_of_one_of_8_ranks", lambda: extra_ransac(S, synth, O, ctx, dev, torch, np, "c4rank8", 16384, 1 << 20, 30, hyp_count=(1 << 20) // 8))
    guarded("match_2048", lambda: extra_match(S, synth, O, ctx, dev, torch, np, 2048, 50))
    guarded("match_4096", lambda: extra_match(S, synth, O, ctx, dev, torch, np, 4096, 50))
    guarded("match_16384", lambda: extra_match(S, synth, O, ctx, dev, torch, np, 16384, 20))
    try:
        out.update(extra_dino(S, O, ctx, dev, torch, np))
    except Exception as e:                                # noqa: BLE001
        out["dino"] = {"error": f"{type(e).__name__}: {e}"}
    for name, v in out.items():                           # the deferred oracle checks
        chk = v.pop("_check", None) if isinstance(v, dict) else None
        if chk is not None:
            try:
                chk()
            except Exception as e:                        # noqa: BLE001
                v["error"] = f"parity check: {type(e).__name__}: {e}"
    return out


# ------------------------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------------------------
def dist_backend():
    """"nccl" (= RCCL), or what SFM_BENCH_DIST_BACKEND names: tests run two ranks on ONE GPU with gloo as torch.distributed's
    backend and tests/fake_ccl behind libsfm_amd_rccl's entry points (tests/test_gpu_fakeccl.py) -- two RCCL ranks cannot share a GPU."""
    return os.environ.get("SFM_BENCH_DIST_BACKEND", "nccl")


def init_dist(dist, rank, world, dev):
    if dist_backend() == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(dist_backend(), rank=rank, world_size=world)


def ctl_device(dev):
    return dev if dist_backend() == "nccl" else "cpu"


def under_profiler():
    """rocprofv3 (or another HSA / HIP tool) is preloaded into this process: child processes would inherit it."""
    return any(k.startswith("ROCP") for k in os.environ) or bool(os.environ.get("HSA_TOOLS_LIB")) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def exchange_probe(S, torch, comm, ctx, pair, n, H, kernel, sweeps, seed, world, fence):
    """100 x (8-byte all-reduce of the key + finalize) on the context stream, back to back: the part of a sharded call that does not
    shrink with the shard."""
    xp = S.default_params(n, num_hypotheses=H, kernel=kernel)
    xp.jacobi_sweeps = sweeps
    xp.seed = seed
    comm.estimate_E(pair, xp)                             # a scored pair
    for _ in range(10):
        comm.exchange_only(pair, xp)
    fence()
    x0 = time.perf_counter()
    for _ in range(100):
        comm.exchange_only(pair, xp)
    fence()
    return {"calls": 100, "seconds": time.perf_counter() - x0, "ranks": world,
            "what": "ncclAllReduce(max, u64) of the key + ransac_finalize_block, on the context stream, back to back (sfm_comm_exchange_only)"}


def exchange_probe_main(args):
    """`bench.py --exchange-probe`: the child process of a one-GPU run (see rank_main)."""
    import numpy as np
    import torch
    import cuda_sfm_amd as S
    from cuda_sfm_amd import synth
    n = args.matches if args.matches is not None else CONFIGS[args.config][0]
    H = args.hyps if args.hyps is not None else CONFIGS[args.config][1]
    dev = torch.device("cuda", 0)
    scene = synth.two_view_scene(n)
    ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.fillXU(torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev))
    comm = S.Comm(ctx, S.Comm.unique_id(), 0, 1)
    out = exchange_probe(S, torch, comm, ctx, pair, n, H, args.kernel, max(args.sweeps, 0), 1, 1, torch.cuda.synchronize)
    out["note"] = "one-rank communicator in a child process: a floor (no peer to wait for)"
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(out), flush=True)
    comm.close()
    return 0


ASSUMED_EXCHANGE_US = 20.0      # a small-message RCCL all-reduce over xGMI + the finalize behind it, as SURVEY.md 8(e) budgets it (10-20 us)


def projection(out, share, c4, c4_share):
    """What ONE GPU can say about 8: this invocation's full-size step against one rank's 1/8 shard, in both step modes, with and
    without the exchange (measured through a one-rank communicator -- a floor -- and at the assumed 20 us)."""
    def ratios(full_ms, share_ms):
        if not full_ms or not share_ms:
            return None
        x_meas = (out.get("exchange_us") or 0.0) / 1e3
        return {"one_gpu_ms_per_step": full_ms, "one_of_8_ranks_ms_per_step": share_ms,
                "excl_exchange": full_ms / share_ms,
                "incl_exchange_as_measured_with_one_rank": full_ms / (share_ms + x_meas),
                "incl_exchange_assumed_20us": full_ms / (share_ms + ASSUMED_EXCHANGE_US / 1e3)}
    p = {"gpus": 8,
         "pipelined": ratios(out["ms_per_step"] if out["config"]["step_mode"].startswith("two-slot") else None, share.get("ms_per_step")),
         "serial": ratios(out.get("serial_ms_per_step"), share.get("serial_ms_per_step")),
         "exchange_us_measured_one_rank": out.get("exchange_us"), "exchange_us_assumed": ASSUMED_EXCHANGE_US,
         "configs3_16384_matches": {"pipelined": ratios(c4.get("ms_per_step"), c4_share.get("ms_per_step")),
                                    "serial": ratios(c4.get("serial_ms_per_step"), c4_share.get("serial_ms_per_step"))},
         "note": "pipelined: back-to-back calls, the exchange of step k overlaps step k + 1 (throughput); serial: ONE call at a time, the "
                 "exchange on the critical path (latency).  In pipelined mode the exchange is hidden unless it exceeds the share's step.",
         "status": "UNMEASURED ON HARDWARE: projections from one GPU running one rank's 1/8 shard in the same invocation; no cross-rank skew, "
                   "no xGMI.  No multi-GPU node has run this code; the driver's SCALE run is the measurement."}
    if p["pipelined"]:
        p["projected_speedup"] = p["pipelined"]["excl_exchange"]          # (the field earlier rounds reported)
    return p


def rank_main(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    if any(args.reserved):              # A/B switches exist only in the lab-bench flavour of the library (make ab; profiles/)
        import cuda_sfm_amd_ab as S
        from cuda_sfm_amd_ab import synth
    else:
        import cuda_sfm_amd as S
        from cuda_sfm_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (there is no CPU fallback)", file=sys.stderr)
        return 2
    if dist_backend() != "nccl":                          # (tests: several ranks on the GPUs there are, tests/test_gpu_fakeccl.py)
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    stage(f"libraries imported, device cuda:{local}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        init_dist(dist, rank, world, dev)
        stage(f"process group up ({dist_backend()})")
    cdev = ctl_device(dev)                                # where the few control-plane tensors of this script live

    n = args.matches if args.matches is not None else CONFIGS[args.config][0]
    H = args.hyps if args.hyps is not None else CONFIGS[args.config][1]
    scene = synth.two_view_scene(n)                       # same bytes on every rank
    d_sift = torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev)
    ctx = S.Context(local, torch.cuda.current_stream().cuda_stream)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.fillXU(d_sift)
    params = S.default_params(n, num_hypotheses=H, kernel=args.kernel)
    for i, v in enumerate(args.reserved[:4]):
        params.reserved[i] = v
    if args.sweeps >= 0:
        params.jacobi_sweeps = args.sweeps
    key_t = torch.zeros(1, dtype=torch.int64, device=dev)

    def reduce_max(t):
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)      # RCCL over xGMI, 8 bytes

    comm = None
    mode = args.comm if world > 1 else "none"
    if args.serial and mode == "rccl":
        mode = "rccl-serial"
    pipelined = (mode == "rccl") or (mode == "none" and not args.serial)
    comm_note = None
    if mode in ("rccl", "rccl-serial"):
        # the C-level exchange (libsfm_amd_rccl.so).  Should its communicator fail to come up on ANY rank, every rank falls
        # back to torch.distributed's all-reduce for the same 8-byte key (agreed through one collective), and the line says so.
        err = ""
        try:
            uid = [S.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)        # the out-of-band hand-over of the ncclUniqueId
            stage("unique id received, ncclCommInitRank next")
            comm = S.Comm(ctx, uid[0], rank, world)
            stage(f"communicator up (nccl_ranks {comm.nccl_ranks()})")
        except Exception as e:                            # noqa: BLE001 -- any failure means "use the other exchange"
            err = f"{type(e).__name__}: {e}"
            stage(f"communicator FAILED: {err}")
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=cdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            comm = None
            comm_note = f"libsfm_amd_rccl communicator unavailable on some rank ({err or 'another rank'}): torch.distributed all-reduce instead"
            print("bench.py: " + comm_note, file=sys.stderr)
            mode = "torch"
            pipelined = False

    # Every step scores NEW hypotheses: step k samples with seed0 + k (the same sequence on every rank), so a result cached from
    # an earlier step could not pass for the current one; the parity check below re-runs and sweeps the LAST timed step's seed.
    seed0 = int(params.seed)
    nstep = [0]

    def step(serial=False):
        nstep[0] += 1
        params.seed = (seed0 + nstep[0]) & 0xFFFFFFFF
        if mode == "rccl" and not serial:
            comm.estimate_E_pipelined(pair, params)
        elif mode in ("rccl", "rccl-serial"):
            comm.estimate_E(pair, params)
        elif mode == "torch":
            S.estimate_E_distributed(pair, params, rank, world, key_t, reduce_max)
        elif pipelined and not serial:
            pair.estimateE_pipelined(params)              # two slots: consecutive steps overlap on the device
        else:
            pair.estimateE(params)

    def fence():
        if comm is not None:
            comm.flush()                                  # the context stream waits for the exchange stream's last finalize
        elif pipelined:
            pair.flush()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: wake the device up (first launches, buffer growth, clock ramp: ~50 ms of work -- counted in time, not in steps: a
    # rank's share of a sharded run is a 0.1 ms step, and twenty of them end before the clock has moved), then the W warm-up
    # steps of the contract
    est_step_s = (H / world) * n / 7e12 + 20e-6           # the same number on every rank: the steps contain a collective
    wake_steps = min(2000, max(20 - args.warmup, int(0.06 / est_step_s)))
    for k in range(max(0, wake_steps)):
        step()
        if k == 0:
            fence()
            stage("first step done")
    for _ in range(args.warmup):
        step()
    fence()
    stage("warm-up done, timed region next")
    # Three hipEventRecord per call cost 3-4 us of a 0.1 ms step (a rank's share of a sharded run) and 0.5-1 % of the headline
    # step (profiles/r03_timed_events_ab.txt); with pipelined steps the kernels overlap inside the timed region anyway and are sampled by serial launches right
    # after it (below), so the events stay out of the timed region unless the steps are serial (or --timed-events on).
    events_in_region = args.timed_events == "on" or (args.timed_events == "auto" and not pipelined)
    if events_in_region:
        ctx.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    solve_ms, score_ms, calls = ctx.kernel_timing_read() if events_in_region else (0.0, 0.0, 0)
    ctx.kernel_timing(False)
    # The contractual region above is ONE sample (12 ms at --steps 20).  Four more regions of the same K steps, timed the same way
    # (fence on both sides, no events), say how much one sample is worth: the line carries median / min / max next to it.
    region_s = [elapsed]
    for _ in range(0 if args.regions <= 1 else args.regions - 1):
        r0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        region_s.append(time.perf_counter() - r0)
    timed_region_kernel_ms = (solve_ms / max(calls, 1), score_ms / max(calls, 1))
    stage("timed regions done")
    # what the timed steps left behind (every rank): winner, E, mask -- of the LAST timed step, whose seed this is
    last_seed = int(params.seed)
    hyp, cnt = pair.get_best()
    main_mask = pair.get_inlier_mask().copy()
    main_E = pair.get_E().copy()
    mask_sum = int(main_mask.sum())
    if pipelined:
        # In the timed region consecutive steps overlap on the device, which stretches every kernel's own duration (two
        # launches share the CUs).  The roofline figures describe the kernel, so they come from serial steps: 20 calls of
        # the one-at-a-time entry point right after the timed region, this rank's shard, no exchange.
        rp = S.default_params(n, num_hypotheses=H, kernel=args.kernel)
        rp.jacobi_sweeps = params.jacobi_sweeps
        for i, v in enumerate(args.reserved[:4]):
            rp.reserved[i] = v
        rp.hyp_begin, rp.hyp_count = S.shard_range(H, rank, world)
        for _ in range(3):
            pair.ransac_score(rp)
        ctx.synchronize()
        ctx.kernel_timing(True)
        for _ in range(20):
            pair.ransac_score(rp)
        solve_ms, score_ms, calls = ctx.kernel_timing_read()
        ctx.kernel_timing(False)
    clock_mhz = pair.last_clock_mhz()
    # Two more regions, so that the line shows what ONE call at a time costs and what the exchange costs on its own:
    #   serial      the same steps through the one-call-at-a-time entry point (the reference's contract: one estimateE per pair,
    #               SfM/sfm.cu:94-153) -- unless the main region already was serial;
    #   exchange    world > 1: the 8-byte all-reduce + the finalize behind it, alone, 100 times (what does not shrink with the shard);
    #               world == 1: the same through a one-rank RCCL communicator (a floor: no peer to wait for).
    other_mode = None
    if pipelined and args.regions > 0:
        ks = max(5, min(args.steps, 50))
        for _ in range(3):
            step(serial=True)
        fence()
        s0 = time.perf_counter()
        for _ in range(ks):
            step(serial=True)
        fence()
        other_mode = {"mode": "serial", "steps": ks, "seconds": time.perf_counter() - s0}
    xprobe = None
    try:
        if comm is not None:
            xprobe = exchange_probe(S, torch, comm, ctx, pair, n, H, args.kernel, params.jacobi_sweeps, last_seed, world, fence)
        elif world == 1 and not any(args.reserved) and args.regions > 0 and not args.no_exchange_probe and not under_profiler():
            # one rank: the same probe through a one-rank RCCL communicator, in a CHILD process -- the one-GPU line must not depend on
            # RCCL coming up (3-4 s, and a library this process otherwise never loads).  Never under a profiler (its preloaded tool would
            # follow the child, whose launches would be averaged into the parent's kernel summaries) -- and the child's environment
            # carries none of the profiler's variables either way.
            stage("exchange probe (child process) next")
            child_env = {k: v for k, v in os.environ.items() if not (k.startswith("ROCP") or k.startswith("ROCPROF") or k in ("LD_PRELOAD", "HSA_TOOLS_LIB"))}
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--exchange-probe", "--matches", str(n), "--hyps", str(H), "--kernel", str(args.kernel),
                                "--sweeps", str(params.jacobi_sweeps)], capture_output=True, text=True, timeout=180, env=child_env)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            xprobe = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": f"child exited {r.returncode}: {r.stderr[-300:]}"}
            stage("exchange probe done")
    except Exception as e:                                # noqa: BLE001 -- a diagnostic: never takes the line down
        xprobe = {"error": f"{type(e).__name__}: {e}"}
    params.seed = last_seed
    per_rank = [[solve_ms / max(calls, 1), score_ms / max(calls, 1), clock_mhz]]
    rc = 0
    agree = None
    if world > 1:
        extra_s = [other_mode["seconds"] if other_mode else 0.0, xprobe["seconds"] if (xprobe and "seconds" in xprobe) else 0.0]
        t = torch.tensor(region_s + extra_s, dtype=torch.float64, device=ctl_device(dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allv = [float(v) for v in t.cpu().numpy()]
        region_s = allv[:len(region_s)]
        if other_mode:
            other_mode["seconds"] = allv[-2]
        if xprobe and "seconds" in xprobe:
            xprobe["seconds"] = allv[-1]
        elapsed = region_s[0]
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
        # every rank must hold the SAME result after the exchange, and the communicator must span all N ranks
        mine = {"rank": rank, "best": [hyp, cnt], "E_sha": hashlib.sha256(main_E.tobytes()).hexdigest()[:16],
                "mask_sha": hashlib.sha256(main_mask.tobytes()).hexdigest()[:16],
                "nccl_ranks": comm.nccl_ranks() if comm is not None else (world if mode == "torch" else 1)}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        same = all((a["best"], a["E_sha"], a["mask_sha"]) == (allr[0]["best"], allr[0]["E_sha"], allr[0]["mask_sha"]) for a in allr)
        spans = all(a["nccl_ranks"] == world for a in allr)
        agree = {"ranks_agree_on_winner_E_mask": bool(same), "communicator_spans_all_ranks": bool(spans), "per_rank": allr}
        if not (same and spans):
            print(f"bench.py: rank results differ or the communicator is short: {json.dumps(allr)}", file=sys.stderr)
            rc = 1

    # one serial call of this rank's own path for the parity check (N = 1: counts of all H hypotheses, key, E, mask)
    serial_result = None
    if world == 1:
        pair.estimateE(params)
        serial_result = (pair.get_inlier_counts(H).copy(), pair.get_key(), pair.get_E().copy(), pair.get_inlier_mask().copy())
        s_hyp, s_cnt = pair.get_best()
        if (s_hyp, s_cnt) != (hyp, cnt) or not np.array_equal(serial_result[2], main_E) or not np.array_equal(serial_result[3], main_mask):
            print("bench.py: the timed (pipelined) steps and a serial estimateE disagree", file=sys.stderr)
            rc = 1
    launch = pair.last_launch()

    # the other null-vector solver, same workload, a short run after the timed region (all ranks take part in its
    # collectives); reported next to the headline, never instead of it
    variant = None
    if not args.no_variants:
        main_sweeps = params.jacobi_sweeps
        params.jacobi_sweeps = 7 if main_sweeps == 0 else 0
        VSTEPS = 20
        for _ in range(3):
            step()
        fence()
        vt0 = time.perf_counter()
        for _ in range(VSTEPS):
            step()
        fence()
        vel = time.perf_counter() - vt0
        if world > 1:
            t = torch.tensor([vel], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            vel = float(t.item())
        vh, vc = pair.get_best()
        variant = {"solver": "normal equations + 7 Jacobi sweeps" if params.jacobi_sweeps == 7 else "householder QR of the 8x9 system",
                   "value": H * VSTEPS / vel, "ms_per_step": 1e3 * vel / VSTEPS, "steps": VSTEPS, "best_hypothesis": vh, "inliers": vc}
        params.jacobi_sweeps = main_sweeps
        params.seed = last_seed

    if rank == 0:
        local_hyps = S.shard_range(H, rank, world)[1]
        score_s = score_ms / 1e3 / max(calls, 1)
        solve_s = solve_ms / 1e3 / max(calls, 1)
        step_mode = ("two-slot pipelined calls: step k + 1 is solved and scored on a second stream and buffer set while step k is still "
                     "running (sfm_estimate_E_pipelined / sfm_estimate_E_sharded_pipelined); --serial times one call at a time"
                     if pipelined else "serial: one estimateE at a time")
        exchange = {"none": "none",
                    "rccl": "ncclAllReduce(max, u64) + finalize on the communicator's exchange stream, overlapped with the next step's solve + scoring (libsfm_amd_rccl.so, sfm_estimate_E_sharded_pipelined)",
                    "rccl-serial": "ncclAllReduce(max, u64) on the compute stream (libsfm_amd_rccl.so, sfm_estimate_E_sharded)",
                    "torch": "torch.distributed all_reduce(MAX), 8 bytes"}[mode]
        if comm_note:
            exchange += " -- " + comm_note
        kname = KERNEL_NAMES.get(launch["kernel"], "?")
        measured_in = (("20 serial launches of this rank's shard right after the timed region (in the timed region consecutive "
                        "steps overlap and stretch each kernel's own duration"
                        + (": solve %.4f ms, scoring %.4f ms per launch there)" % timed_region_kernel_ms if events_in_region else
                           "; no events are recorded there: three hipEventRecord per call cost 3-4 us per step, --timed-events on samples them)"))
                       if pipelined else "the timed region")
        out = {
            "metric": "RANSAC E-matrix hypotheses/sec (8-point, fused scoring), inlier-mask parity vs CPU oracle",
            "value": H * args.steps / elapsed,
            "unit": "hypotheses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_regions": regions_summary(region_s, args.steps),
            "serial_ms_per_step": (1e3 * other_mode["seconds"] / other_mode["steps"]) if other_mode else (1e3 * elapsed / args.steps if not pipelined else None),
            "exchange_us": (1e6 * xprobe["seconds"] / xprobe["calls"]) if (xprobe and "seconds" in xprobe) else None,
            "exchange_probe": xprobe,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"synthetic two-view scene, {n} matches (30% outliers, 0.5 px noise), "
                                   f"{H} 8-point hypotheses per step sharded over {world} GPU(s), estimateE end to end "
                                   "(sample+solve+score+argmax+winner E+inlier mask)",
                       "preset": args.config if (args.matches is None and args.hyps is None) else "custom",
                       "preset_note": CONFIGS[args.config][2] if (args.matches is None and args.hyps is None) else None,
                       "matches": n, "hypotheses_per_step": H, "threshold": params.threshold,
                       "sampler_seed": f"step k samples with seed {seed0} + k (every step scores new hypotheses); the parity check sweeps the last timed step's seed {last_seed}",
                       "jacobi_sweeps": params.jacobi_sweeps,
                       "solver": "householder QR of the 8x9 system" if params.jacobi_sweeps == 0 else f"normal equations + {params.jacobi_sweeps} Jacobi sweeps",
                       "kernel": dict(launch, name=kname), "step_mode": step_mode, "exchange": exchange,
                       "untimed_steps_before_the_timed_region": {"device_wake_up": max(0, wake_steps), "warmup": args.warmup,
                                                                 "note": "the wake-up covers ~60 ms of work whatever the step length (clock ramp); neither is timed"},
                       "hip_events_inside_the_timed_region": bool(events_in_region),
                       "nccl_ranks": comm.nccl_ranks() if comm is not None else (world if mode == "torch" else 1),
                       "per_rank_kernel_ms": [{"rank": r, "solve": v[0], "score": v[1], "shader_clock_mhz": v[2]} for r, v in enumerate(per_rank)]},
            "roofline": roofline_block(launch["kernel"], n, local_hyps, score_s, solve_s, clock_mhz, measured_in,
                                       rule=reserved_rule(args.reserved)),
            "result": {"best_hypothesis": hyp, "inliers": cnt, "mask_sum": mask_sum, "sampler_seed_of_this_result": last_seed},
        }
        # which form of the pre-filter the timed steps ran (the first call after a fillXU runs per-hypothesis operands; the wake-up and warm-up
        # steps come before the timed region, so the timed steps run per-tile operands): sfm_ransac_last_prefilter_rule
        out["roofline"]["operands"] = {2: "per hypothesis (whole-view boxes)", 3: "per (hypothesis, tile) over the ordered copy"}.get(launch.get("prefilter_rule"))
        if agree is not None:
            out["result"]["multi_gpu"] = agree
        if variant is not None:
            out["variants"] = [variant]
        O = None
        if world == 1 and not args.no_cpu:
            O = load_oracle()
            _, _, X0, X1 = O.fill_xu(scene["sift"], scene["Kinv"])
            base, okey, ocounts = cpu_baseline(O, X0, X1, params, n, H, args.cpu_seconds, scene)
            out["cpu_baseline"] = base
            ok = full_parity(O, X0, X1, params, n, okey, ocounts, serial_result)
            out["result"]["parity_vs_oracle"] = ok
            out["result"]["parity_checked"] = (f"all {H} inlier counts, the arg-max key (first maximum), the winner's E bit for bit and the inlier mask "
                                               f"of the LAST timed step's hypotheses (sampler seed {last_seed}) against the CPU oracle: the step's own "
                                               "winner / E / mask as the timed region left them, and every count from one serial estimateE with that seed")
            if not ok:
                rc = 1
        if world == 1 and not args.no_extra:
            skip = {args.config} if (args.matches is None and args.hyps is None) else set()
            out["extra"] = run_extras(S, synth, O, ctx, dev, torch, np, skip)
            share = out["extra"].get("headline_share_of_one_of_8_ranks", {})
            if args.config == "headline" and args.matches is None and args.hyps is None and share.get("ms_per_step"):
                out["scaling_projection"] = projection(out, share, out["extra"].get("c4_one_gpu", {}), out["extra"].get("c4_share_of_one_of_8_ranks", {}))
            bad = [k for k, v in out["extra"].items() if isinstance(v, dict) and (v.get("parity_vs_oracle") is False or "error" in v)]
            if bad:
                print(f"bench.py: extra runs failed or lost parity: {bad}", file=sys.stderr)
                rc = 1
        try:                                        # RCCL writes a start-up banner through C stdio: push it out BEFORE the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


# ------------------------------------------------------------------------------------------------------------------
# --config c5: BASELINE configs[4] as a job of its own (N >= 1 ranks)
# ------------------------------------------------------------------------------------------------------------------
def c5_rank_main(args):
    """One "step" = the whole many-views job: 36 host images (the reference's dino frames, 8-bit grey fixtures) -> ExtractSift per
    view on the rank that owns it -> the count-sized feature exchange -> MatchSiftData + the Image_pair sequence for every pair on
    the rank that owns it -> one gather of the result records.  PCIe-inclusive by construction (the job starts from host images).
    N > 1: --comm rccl = sfm_process_views_sharded (C, RCCL), --comm torch = the Python harness over torch.distributed."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import cuda_sfm_amd as S
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import read_pnm_grey, dino_frame, DINO_K, DINO_KINV, DINO_SIFT

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (there is no CPU fallback)", file=sys.stderr)
        return 2
    if not os.path.exists(dino_frame(35)):
        print("bench.py --config c5: tests/golden/dino fixtures not present", file=sys.stderr)
        return 2
    if dist_backend() != "nccl":                          # (tests: several ranks on the GPUs there are, tests/test_gpu_fakeccl.py)
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    stage(f"libraries imported, device cuda:{local}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        init_dist(dist, rank, world, dev)
        stage(f"process group up ({dist_backend()})")
    views_f = [read_pnm_grey(dino_frame(k)) for k in range(36)]
    views8 = [v.astype(np.uint8) for v in views_f]
    pairs = S.ring_pairs(36) if args.pairs == "ring" else [(i, j) for i in range(36) for j in range(i + 1, 36)]
    ctx = S.Context(local, torch.cuda.current_stream().cuda_stream)
    mode = "none" if world == 1 else ("torch" if args.comm == "torch" else "rccl")
    comm = None
    comm_note = None
    xstats = {}

    def gather_results(t):
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out

    def torch_step():
        return S.process_views(ctx, views8, DINO_K, DINO_KINV, pairs=pairs, rank=rank, world=world, max_pts=8192, sift=DINO_SIFT,
                               dist=dist if world > 1 else None, gather_results=gather_results if world > 1 else None, device=dev, stats=xstats)

    if mode == "rccl":
        # The C path (sfm_process_views_sharded_u8: ~36 grouped ncclBroadcasts, two all-gathers, two status all-reduces) has only ever
        # run with more than one rank over the shared-memory stand-in of tests/fake_ccl.  Should the communicator or the first,
        # untimed job fail on ANY rank, every rank falls back to the torch.distributed harness (agreed through one collective) and
        # the line says so; when both paths work, their records are compared once.
        err = ""
        first = None
        try:
            uid = [S.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            stage("unique id received, ncclCommInitRank next")
            comm = S.Comm(ctx, uid[0], rank, world)
            stage(f"communicator up (nccl_ranks {comm.nccl_ranks()})")
            first = comm.process_views(views8, DINO_K, DINO_KINV, pairs=pairs, max_pts=8192, sift=DINO_SIFT)
            stage("first sharded job done (C path)")
        except Exception as e:                            # noqa: BLE001 -- any failure means "use the other exchange"
            err = f"{type(e).__name__}: {e}"
            stage(f"C path FAILED: {err}")
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=ctl_device(dev))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            comm_note = f"sfm_process_views_sharded unavailable on some rank ({err or 'another rank'}): torch.distributed harness instead"
            print("bench.py: " + comm_note, file=sys.stderr)
            if comm is not None:
                try:
                    comm.close()
                except Exception:                         # noqa: BLE001
                    pass
            comm = None
            mode = "torch"
        else:
            tres, tcounts = torch_step()                  # cross-check of the first real multi-rank run: same records from both paths
            same = list(tcounts) == list(first[1]) and sorted(tres) == sorted(first[0]) and all(np.array_equal(tres[k].view(np.uint32), first[0][k].view(np.uint32)) for k in tres)
            xstats["c_path_equals_torch_path"] = bool(same)
            stage(f"C path against the torch harness: {'identical records' if same else 'RECORDS DIFFER'}")

    def step():
        if comm is not None:
            res, counts = comm.process_views(views8, DINO_K, DINO_KINV, pairs=pairs, max_pts=8192, sift=DINO_SIFT)
            xstats["feature_bytes"], xstats["slot_bytes"] = comm.last_exchange()
            return res, counts
        return torch_step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: the HIP runtime grows an internal pool once, ~15000 launches into a process (a 40 ms step near call 25 of this job,
    # profiles/c5_step_probe.py): enough untimed steps to be past it whatever --warmup says, then the W warm-up steps of the contract
    wake_steps = max(0, 40 - args.warmup)
    for k in range(wake_steps + max(args.warmup, 2)):
        res, counts = step()
        if k == 0:
            stage("first step done")
    fence()
    stage("warm-up done, timed regions next")
    steps = args.steps
    region_s = []
    for _ in range(max(1, args.regions)):
        t0 = time.perf_counter()
        for _ in range(steps):
            res, counts = step()
        fence()
        region_s.append(time.perf_counter() - t0)
    stage("timed regions done")
    rc = 0
    if xstats.get("c_path_equals_torch_path") is False:
        print("bench.py: sfm_process_views_sharded and the torch.distributed harness returned different records", file=sys.stderr)
        rc = 1
    rec = np.stack([res[k] if k in res else np.full(28, -1.0, np.float32) for k in range(len(pairs))])
    agree = None
    if world > 1:
        t = torch.tensor(region_s, dtype=torch.float64, device=ctl_device(dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_s = [float(v) for v in t.cpu().numpy()]
        mine = {"rank": rank, "records_sha": hashlib.sha256(rec.tobytes()).hexdigest()[:16], "pairs_done": len(res), "counts_sha": hashlib.sha256(np.asarray(counts, np.int32).tobytes()).hexdigest()[:16],
                "nccl_ranks": comm.nccl_ranks() if comm is not None else world, "exchange": dict(xstats)}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        same = all((a["records_sha"], a["counts_sha"], a["pairs_done"]) == (allr[0]["records_sha"], allr[0]["counts_sha"], allr[0]["pairs_done"]) for a in allr)
        spans = all(a["nccl_ranks"] == world for a in allr)
        agree = {"ranks_hold_identical_records": bool(same), "communicator_spans_all_ranks": bool(spans), "per_rank": allr}
        if not (same and spans):
            print(f"bench.py: rank results differ or the communicator is short: {json.dumps(allr)}", file=sys.stderr)
            rc = 1
    if rank == 0:
        elapsed = region_s[0]
        real = int(sum(counts)) * 576
        out = {"metric": "view pairs/sec: MatchSiftData + estimateE + pose + triangulation per pair, ExtractSift per view, host images in (BASELINE configs[4])",
               "value": len(pairs) * steps / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": max(args.warmup, 2), "untimed_wake_up_steps": wake_steps,
               "ms_per_step": 1e3 * elapsed / steps, "ms_per_step_regions": regions_summary(region_s, steps), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "the reference's 36 dino frames (data/dino/viff.000-035.ppm as 8-bit grey fixtures, tests/golden/dino)",
               "config": {"workload": f"36 views 720 x 576, {len(pairs)} view pairs ({args.pairs}), H = n/8 hypotheses per pair (the reference's own count), views and pairs dealt round-robin over {world} GPU(s)",
                          "preset": "c5", "preset_note": CONFIGS["c5"][2], "features_per_view": [int(min(counts)), int(max(counts))],
                          "path": {"none": "one rank: sfm_extract_views_u8 + sfm_process_pairs (no exchange)",
                                   "rccl": "sfm_process_views_sharded_u8 (libsfm_amd_rccl.so): counts all-gather, count-sized grouped ncclBroadcast of the features, records all-gather",
                                   "torch": "Python harness over torch.distributed (RCCL): counts all_gather, per-view broadcast, records all_gather"}[mode]
                                  + (" -- " + comm_note if comm_note else ""),
                          "host_image_format": "8-bit grey (what cv::imread(path, 0) delivers, src/main.cpp:249), the same on every path and rank count",
                          "c_path_equals_torch_path": xstats.get("c_path_equals_torch_path"),
                          "nccl_ranks": comm.nccl_ranks() if comm is not None else (world if mode == "torch" else 1)},
               "exchange": {"feature_bytes_into_each_rank": xstats.get("feature_bytes", 0), "sum_count_x_576": real,
                            "over_real_bytes": (xstats.get("feature_bytes", 0) / real) if (real and world > 1) else None,
                            "max_pts_slot_bytes_equivalent": xstats.get("slot_bytes", 0),
                            "note": "bytes every rank receives in the feature exchange; before ABI version 2 the exchange moved max_pts-sized slots (4.6 x the real bytes on these frames)"},
               "roofline": None,
               "roofline_note": "a pipeline of ~50 launch-bound kernels per view / pair batch, not one kernel: no roofline is claimed (the headline line carries the scoring kernel's)",
               "scaling_status": ("one GPU" if world == 1 else f"{world} ranks ran") + "; no multi-GPU curve of this configuration had been measured by the builder (1-GPU boxes only): the driver's SCALE run is the measurement",
               "result": {"pairs_done": len(res)}}
        if agree is not None:
            out["result"]["multi_gpu"] = agree
        if world == 1 and not args.no_cpu:
            O = load_oracle()
            d_feats = []
            pitch = (720 + 127) // 128 * 128
            for v in views_f:
                pad = np.zeros((576, pitch), np.float32); pad[:, :720] = v
                d_sift = torch.zeros((8192, 576), dtype=torch.uint8, device=dev)
                nv, _ = ctx.extract_sift(d_sift, 8192, torch.from_numpy(pad).to(dev), 720, 576, pitch, **DINO_SIFT)
                d_feats.append(d_sift.cpu().numpy().view(O.SIFT_DTYPE).reshape(-1)[:nv].copy())
            t0 = time.perf_counter()
            bad = dino_pairs_vs_oracle(S, O, np, d_feats, pairs, res, DINO_KINV)
            dt = time.perf_counter() - t0
            out["result"]["parity_vs_oracle"] = bool(len(res) == len(pairs) and not bad)
            out["result"]["parity_pairs_checked"] = len(pairs)
            out["cpu_baseline"] = {"value": len(pairs) / dt, "unit": "pairs/s", "cores": len(os.sched_getaffinity(0)), "kind": "port",
                                   "sample": f"the oracle chain (match + fillXU + estimateE + poses, no extraction) for all {len(pairs)} pairs, {dt:.1f} s; it is also the parity sweep"}
            if bad:
                out["result"]["pairs_that_differ"] = bad[:8]
                rc = 1
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launcher_main(args, argv)
    if args.exchange_probe:
        return exchange_probe_main(args)
    if args.config == "c5":
        if "--steps" not in " ".join(argv):
            args.steps = 10
        return c5_rank_main(args)
    return rank_main(args)


if __name__ == "__main__":
    sys.exit(main())
