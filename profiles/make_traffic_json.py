"""profiles/pmc_<tag>_headline_summary.txt -> r02_traffic.json (what bench.py quotes as roofline.traffic_profiled /
issue_profiled): HBM KB per launch and issue-slot occupancy of the two RANSAC kernels of the default bench command."""
import json
import re
import sys

text = open(sys.argv[1]).read()
out = {"_comment": "per launch, default bench command (4096 matches, 2^20 hypotheses), rocprofv3 --pmc passes of profiles/collect_r02.sh; "
                   "FETCH_SIZE / WRITE_SIZE in KB (separate passes); busy fractions = quad-cycle counters x 4 over (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs"}
for block in re.split(r"\n(?=sfm::)", text):
    name = block.split("\n", 1)[0].strip().replace("sfm::", "")
    if not name:
        continue
    vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\w+)\s+(\d+) per launch", block, re.M)}
    key = name.split("<")[0]
    if key not in ("ransac_score_prefilter", "ransac_solve_lanes2", "ransac_solve_lanes1_qr", "ransac_score_waves"):
        continue
    e = {"matches": 4096, "hypotheses": 1 << 20, "fetch_kb": vals.get("FETCH_SIZE"), "write_kb": vals.get("WRITE_SIZE"),
         "valu_insts_per_launch": vals.get("SQ_INSTS_VALU")}
    cyc = vals.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc > 0:
        e["kernel_cycles"] = cyc
        if "SQ_ACTIVE_INST_VALU" in vals:
            e["valu_busy_frac"] = round(4.0 * vals["SQ_ACTIVE_INST_VALU"] / (cyc * 1024.0), 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
            e["mfma_busy_frac"] = round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4)
        if vals.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = round(vals.get("SQ_LDS_BANK_CONFLICT", 0.0) / vals["SQ_LDS_IDX_ACTIVE"], 4)
    out[key] = e
print(json.dumps(out, indent=1))
