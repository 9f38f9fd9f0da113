"""gpurun_out/pmc_<tag>_summary.txt (profiles/pmc_pf.sh) -> profiles/r06_traffic.json (what bench.py quotes as roofline.traffic /
issue_profiled): HBM KB per launch and issue-slot occupancy of the RANSAC kernels of the default bench command, together with
the hash of the kernel sources they were collected on (bench.py refuses to quote them for other sources).
usage: python profiles/make_traffic_json.py <summary.txt> [matches hypotheses] > profiles/r06_traffic.json"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

text = open(sys.argv[1]).read()
matches = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
hyps = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
out = {"_comment": "per launch, `bench.py --serial` at %d matches x %d hypotheses, rocprofv3 --pmc passes of profiles/pmc_pf.sh; "
                   "FETCH_SIZE / WRITE_SIZE in KB (separate passes); busy fractions = quad-cycle counters x 4 over (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs" % (matches, hyps),
       "code_sha256_16": bench.source_hash(), "code_files": list(bench.PF_SOURCES)}
for block in re.split(r"\n(?=sfm::)", text):
    name = block.split("\n", 1)[0].strip().replace("sfm::", "")
    if not name:
        continue
    vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\w+)\s+(\d+) per launch", block, re.M)}
    launches = max([int(m.group(1)) for m in re.finditer(r"\((\d+) launches\)", block)] or [0])
    key = name.split("<")[0]
    # the scoring kernel's two forms: the first call after a fillXU runs per-hypothesis operands (<16, 0, 2, ...> with
    # ransac_solve_lanes1_qr_rec), all later steps the tile form (<16, 0, 3, ...>): the latter is what the bench times
    if key == "ransac_score_prefilter" and re.match(r"ransac_score_prefilter<16, 0, 2\b", name):
        key = "ransac_score_prefilter_per_hypothesis_form"
    if key not in ("ransac_score_prefilter", "ransac_score_prefilter_per_hypothesis_form", "ransac_solve_lanes2", "ransac_solve_lanes1_qr", "ransac_solve_lanes1_qr_rec", "ransac_score_waves"):
        continue
    e = {"kernel": name, "launches_profiled": launches, "matches": matches, "hypotheses": hyps, "fetch_kb": vals.get("FETCH_SIZE"), "write_kb": vals.get("WRITE_SIZE"),
         "valu_insts_per_launch": vals.get("SQ_INSTS_VALU"), "salu_insts_per_launch": vals.get("SQ_INSTS_SALU"), "lds_insts_per_launch": vals.get("SQ_INSTS_LDS")}
    cyc = vals.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc > 0:
        e["kernel_cycles"] = cyc
        if "SQ_ACTIVE_INST_VALU" in vals:
            e["valu_busy_frac"] = round(4.0 * vals["SQ_ACTIVE_INST_VALU"] / (cyc * 1024.0), 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
            e["mfma_busy_frac"] = round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4)
        if vals.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = round(vals.get("SQ_LDS_BANK_CONFLICT", 0.0) / vals["SQ_LDS_IDX_ACTIVE"], 4)
        if vals.get("SQ_WAVE_CYCLES"):
            e["mean_resident_waves_per_simd"] = round(4.0 * vals["SQ_WAVE_CYCLES"] / (cyc * 1024.0), 3)
            # where a wavefront's cycles go (disjoint: issuing / parked at s_waitcnt / stalled at issue)
            for k, name in (("SQ_ACTIVE_INST_ANY", "wave_issuing_frac"), ("SQ_WAIT_ANY", "wave_parked_at_waitcnt_frac"), ("SQ_WAIT_INST_ANY", "wave_issue_stalled_frac")):
                if k in vals:
                    e[name] = round(vals[k] / vals["SQ_WAVE_CYCLES"], 4)
        if vals.get("SQ_LDS_IDX_ACTIVE") and cyc > 0:
            e["lds_busy_frac"] = round(vals["SQ_LDS_IDX_ACTIVE"] / (cyc * 256.0), 4)
    out[key] = e
print(json.dumps(out, indent=1))
