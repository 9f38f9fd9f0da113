"""Round 6: what real RCCL does with the exchange step, on ONE GPU -- the hazards the shared-memory stand-in of tests/fake_ccl cannot show.

Runs under  rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --output-format csv -d <dir> -o x -- python3 profiles/exchange_trace.py run
(profiles/r06_exchange_trace.sh) and then, as `python3 profiles/exchange_trace.py report <dir>`, reads the traces:
  1. sfm_estimate_E_sharded_pipelined through a ONE-rank RCCL communicator, 40 steps of a rank's share (4096 matches x 131072 hypotheses):
     does the exchange stream's work of step k (ncclAllReduce -- a device-to-device copy or an RCCL kernel with one rank -- and
     ransac_finalize_block) overlap the solve / scoring kernels of step k + 1 on the compute streams?
  2. sfm_process_views_sharded (6 dino views, all pairs) through the same communicator: is there a host synchronisation
     (hipStreamSynchronize / hipDeviceSynchronize / hipEventSynchronize) between ncclGroupStart and ncclGroupEnd of the grouped
     per-view broadcasts (comm.cpp)?  The HIP calls RCCL makes inside the group are listed.
One rank is what a one-GPU box can run (two RCCL ranks on one device are refused by the runtime): RCCL itself, its streams and its
enqueue path are exercised; peers, xGMI and skew are not."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    import torch
    import cuda_sfm_amd as S
    from cuda_sfm_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dev = torch.device("cuda", 0)
    n, H = 4096, 131072
    scene = synth.two_view_scene(n)
    ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, n)
    pair.fillXU(torch.from_numpy(scene["sift"].view(np.uint8).reshape(n, 576)).to(dev))
    comm = S.Comm(ctx, S.Comm.unique_id(), 0, 1)
    p = S.default_params(n, num_hypotheses=H)
    for k in range(10):
        p.seed = 100 + k
        comm.estimate_E_pipelined(pair, p)
    comm.flush(); torch.cuda.synchronize()
    torch.cuda.nvtx.range_push("pipelined_exchange") if hasattr(torch.cuda, "nvtx") else None
    for k in range(40):
        p.seed = 200 + k
        comm.estimate_E_pipelined(pair, p)
    comm.flush(); torch.cuda.synchronize()
    hyp, cnt = pair.get_best()
    print(json.dumps({"part": 1, "steps": 40, "best": [hyp, cnt], "nccl_ranks": comm.nccl_ranks()}), flush=True)
    # part 2: the count-sized feature exchange of configs[4] (grouped broadcasts)
    from helpers import read_pnm_grey
    dino = os.path.join(ROOT, "tests", "golden", "dino")
    names = sorted(f for f in os.listdir(dino) if f.startswith("dino_grey_"))[:6]
    images = [read_pnm_grey(os.path.join(dino, f)) for f in names]
    pairs = [(i, j) for i in range(len(images)) for j in range(i + 1, len(images))]
    K = np.float32([[2360, 0, 360], [0, 2360, 288], [0, 0, 1]]); Kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    res, counts = comm.process_views(images, K, Kinv, pairs=pairs, max_pts=4096)
    torch.cuda.synchronize()
    print(json.dumps({"part": 2, "views": len(images), "pairs": len(res), "features": counts, "exchange": comm.last_exchange()}), flush=True)
    comm.close()


def load(d, suffix):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def report(d):
    kern = load(d, "kernel_trace.csv")
    api = load(d, "hip_api_trace.csv")
    mem = load(d, "memory_copy_trace.csv")
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in kern))
    print(f"{len(ks)} kernel records, {len(api)} HIP API records, {len(mem)} memory copies")
    by_name = {}
    for s, e, nme, q, st in ks:
        by_name.setdefault(nme, []).append((s, e, q, st))
    for nme, v in sorted(by_name.items(), key=lambda kv: -len(kv[1]))[:12]:
        qs = sorted({x[2] for x in v}); sts = sorted({x[3] for x in v})
        print(f"  {len(v):5d} x {nme[:70]:70s} queues {qs} streams {sts} avg {sum(e - s for s, e, _, _ in v) / len(v) / 1e3:.1f} us")
    # part 1: finalize launches of the LAST 40 pipelined steps against the solve / scoring kernels running at the same time
    fin = [x for x in by_name.get("sfm::ransac_finalize_block", [])]
    score = [x for nme, v in by_name.items() if "ransac_score_prefilter" in nme for x in v]
    solve = [x for nme, v in by_name.items() if "ransac_solve_lanes1_qr" in nme for x in v]
    rccl = [x for nme, v in by_name.items() if "nccl" in nme.lower() for x in v]
    compute = sorted(score + solve)
    fin = sorted(fin)[-45:-5] if len(fin) >= 45 else sorted(fin)
    ov = []
    for s, e, q, st in fin:
        o = sum(max(0, min(e, ce) - max(s, cs)) for cs, ce, cq, cst in compute if (cq, cst) != (q, st))
        ov.append((e - s, o))
    if ov:
        full = sum(1 for d_, o in ov if o >= 0.9 * d_)
        print(f"part 1: {len(ov)} finalize launches on the exchange stream: mean duration {sum(d_ for d_, _ in ov) / len(ov) / 1e3:.2f} us, "
              f"mean time overlapped by solve / scoring kernels of OTHER queues {sum(o for _, o in ov) / len(ov) / 1e3:.2f} us; "
              f"{full} of {len(ov)} run entirely under the next step's kernels")
        print(f"        finalize queue / stream ids {sorted({(q, st) for _, _, q, st in fin})}; scoring {sorted({(q, st) for _, _, q, st in score})}; solve {sorted({(q, st) for _, _, q, st in solve})}")
    print(f"        RCCL kernels in the trace: {len(rccl)} (a one-rank all-reduce of 8 bytes in place is a no-op or a copy: {len(mem)} memory copies)")
    # part 2: the RCCL API trace (--rccl-trace) names every ncclGroupStart / ncclBroadcast / ncclGroupEnd; a host synchronisation of the
    # same thread between a group's start and its end would serialise the grouped broadcasts
    rapi = sorted(load(d, "rccl_api_trace.csv"), key=lambda r: int(r["Start_Timestamp"]))
    calls = {}
    for r in rapi:
        calls[r["Function"]] = calls.get(r["Function"], 0) + 1
    print(f"part 2: RCCL API calls: {dict(sorted(calls.items()))}")
    syncs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id")) for r in api
                   if r["Function"] in ("hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize"))
    groups, open_t = [], {}
    for r in rapi:
        tid = r.get("Thread_Id")
        if r["Function"] == "ncclGroupStart":
            open_t[tid] = [int(r["Start_Timestamp"]), 0, 0]
        elif r["Function"] == "ncclBroadcast" and tid in open_t:
            open_t[tid][2] += 1
        elif r["Function"] == "ncclGroupEnd" and tid in open_t:
            g = open_t.pop(tid); g[1] = int(r["End_Timestamp"]); groups.append((g[0], g[1], g[2], tid))
    worst = 0
    for g0, g1, nb, tid in groups:
        inside = [s_ for s_ in syncs if g0 <= s_[0] <= g1 and s_[3] == tid]
        worst = max(worst, len(inside))
    nb_all = [g[2] for g in groups]
    print(f"        {len(groups)} ncclGroupStart..ncclGroupEnd groups, broadcasts per group {sorted(set(nb_all))}, longest group {max((g[1] - g[0]) for g in groups) / 1e3 if groups else 0:.1f} us; "
          f"host synchronisations of the calling thread INSIDE a group: {worst} at most; {len(syncs)} host synchronisations in the whole run")
    allred = [r for r in rapi if r["Function"] == "ncclAllReduce"]
    if allred:
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in allred]
        print(f"        ncclAllReduce: {len(allred)} calls, host-side enqueue {sum(dur) / len(dur) / 1e3:.1f} us on average (max {max(dur) / 1e3:.1f})")


if __name__ == "__main__":
    if sys.argv[1:2] == ["run"]:
        run()
    else:
        report(sys.argv[2] if len(sys.argv) > 2 else ".")
