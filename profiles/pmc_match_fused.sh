#!/bin/sh
# Counters of the fused matcher at one size (N, default 4096): separate --pmc passes, kernel trace only (no other trace domain).
# sh profiles/pmc_match_fused.sh [n] -> gpurun_out/pmc_match_fused_summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export N=${1:-4096}
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_mf_1 -o p -- python3 $R/profiles/match_fused_probe.py > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pmc_mf_2 -o p -- python3 $R/profiles/match_fused_probe.py > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_mf_f -o p -- python3 $R/profiles/match_fused_probe.py > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_mf_w -o p -- python3 $R/profiles/match_fused_probe.py > /dev/null 2>&1
cd $R
python3 - <<PY > $O/pmc_match_fused_summary.txt
import csv, glob, collections, os
N = int(os.environ["N"])
print("fused matcher, %d x %d x 128, per launch (averages over the launches of profiles/match_fused_probe.py)" % (N, N))
dur = []
for f in glob.glob("$O/pmc_mf_1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "match_fused" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
avg_ns = sum(dur) / max(1, len(dur))
print("kernel duration under counters: %.1f us (%d launches)" % (avg_ns / 1e3, len(dur)))
tot = collections.defaultdict(float); cnt = collections.Counter()
for d in ("pmc_mf_1", "pmc_mf_2", "pmc_mf_f", "pmc_mf_w"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "match_fused" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
# one CSV row per (dispatch, counter): values are already summed over the XCDs / SEs by rocprofv3
for k in sorted(tot):
    print("%-32s %16.0f" % (k, tot[k] / max(1, cnt[k])))
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot and avg_ns > 0:
    fkb, wkb = tot["FETCH_SIZE"] / cnt["FETCH_SIZE"], tot["WRITE_SIZE"] / cnt["WRITE_SIZE"]
    print("HBM traffic per launch: fetch %.1f MB, write %.1f MB (FETCH_SIZE / WRITE_SIZE in KB, separate passes) -> %.0f GB/s over the kernel's duration; algorithmic: 2 x %d rows x 512 B = %.1f MB read once"
          % (fkb / 1e3, wkb / 1e3, (fkb + wkb) * 1e3 / avg_ns * 1e3 / 1e3, N, 2 * N * 512 / 1e6))
PY
cat $O/pmc_match_fused_summary.txt
