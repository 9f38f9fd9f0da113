"""Timeline of ONE sfm_extract_sift call out of a rocprofv3 kernel trace of profiles/sift_bench.py:
   python profiles/sift_timeline.py gpurun_out/r01_sstats/sift_kernel_trace.csv > profiles/r01_sift_timeline_1080p.json
Takes the last call on the largest image (the dispatch run lowpass ... desc whose low-pass grid is the largest)."""
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sfm::sift_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
calls, cur = [], []
for r in rows:
    if "sift_lowpass_kernel" in r["Kernel_Name"] and cur:
        calls.append(cur); cur = []
    cur.append(r)
calls.append(cur)
big = max(int(c[0]["Grid_Size_X"]) for c in calls)
call = [c for c in calls if int(c[0]["Grid_Size_X"]) == big][-1]
t0 = int(call[0]["Start_Timestamp"])
out = {"what": "one sfm_extract_sift call on the largest image of profiles/sift_bench.py (1920x1080 synthetic, 5 octaves, thresh 3.0), "
               "rocprofv3 --kernel-trace", "dispatches": []}
for r in call:
    out["dispatches"].append({"kernel": r["Kernel_Name"].split("(")[0].replace("sfm::", ""),
                              "start_us": (int(r["Start_Timestamp"]) - t0) / 1e3,
                              "duration_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                              "grid": int(r["Grid_Size_X"]), "workgroup": int(r["Workgroup_Size_X"]),
                              "lds_bytes": int(r["LDS_Block_Size"]), "vgprs": int(r["VGPR_Count"])})
out["gpu_span_us"] = (int(call[-1]["End_Timestamp"]) - t0) / 1e3
print(json.dumps(out, indent=1))
