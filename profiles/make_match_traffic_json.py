"""gpurun_out/pmc_r06_match_<n>_summary.txt (profiles/pmc_match.sh) -> profiles/r06_match_traffic.json: HBM KB per launch (FETCH_SIZE / WRITE_SIZE,
separate --pmc passes) and matrix-pipe / vector occupancy of the matcher kernels per size; bench.py quotes it in extra.match_<n>.traffic together
with the GB/s at the time it measures (the north_star's "rocprof HBM GB/s on the match kernel").
usage: python profiles/make_match_traffic_json.py <n>:<summary.txt> [<n>:<summary.txt> ...] > profiles/r06_match_traffic.json"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MATCH_SOURCES = ("match.hip", "match_fused.hip", "match_prefilter.hip", "match_common.hpp", "match_prefilter_math.hpp")


def source_hash():
    h = hashlib.sha256()
    for name in MATCH_SOURCES:
        with open(os.path.join(ROOT, "cuda-sfm_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def demangle(name):
    # the kernels that take _Float16 pointers defeat c++filt (DF16_): read the name and the integer template arguments directly
    m = re.match(r"_ZN3sfm(\d+)", name)
    if m:
        start = m.end()
        base = name[start:start + int(m.group(1))]
        rest = name[start + int(m.group(1)):]
        targs = ""
        if rest.startswith("I"):
            targs = "<" + ", ".join(re.findall(r"Li(\d+)E", rest[:rest.index("EEv") + 1] if "EEv" in rest else rest)) + ">"
        return base + targs
    name = name.replace("void ", "").replace("sfm::", "")
    return name.split("(")[0]


if __name__ == "__main__":
    out = {"_comment": "per launch of each matcher kernel of sfm_match_soa at n x n descriptors, rocprofv3 --pmc passes of profiles/pmc_match.sh; FETCH_SIZE / WRITE_SIZE in KB",
           "code_sha256_16": source_hash(), "code_files": list(MATCH_SOURCES), "sizes": {}}
    for arg in sys.argv[1:]:
        n, path = arg.split(":", 1)
        text = open(path).read()
        kernels = {}
        name = None
        vals = {}

        def close():
            if name is None or "match" not in name:
                return
            e = {"fetch_kb": vals.get("FETCH_SIZE"), "write_kb": vals.get("WRITE_SIZE"), "valu_insts": vals.get("SQ_INSTS_VALU"),
                 "mfma_busy_cycles": vals.get("SQ_VALU_MFMA_BUSY_CYCLES"), "lds_bank_conflict_cycles": vals.get("SQ_LDS_BANK_CONFLICT"),
                 "lds_active_cycles": vals.get("SQ_LDS_IDX_ACTIVE")}
            cyc = vals.get("GRBM_GUI_ACTIVE", 0.0) / 8.0       # the counter sums the 8 XCDs
            if cyc > 0:
                e["gpu_active_cycles_under_counters"] = cyc
            kernels[demangle(name)] = e

        for line in text.splitlines():
            if not line.strip() or "avg ns under counters" in line or line.lstrip().startswith("->"):
                continue
            if not line[0].isspace():
                close()
                name, vals = line.strip(), {}
                continue
            m = re.match(r"\s+(\w+)\s+(\d+) per launch", line)
            if m:
                vals[m.group(1)] = float(m.group(2))
        close()
        out["sizes"][n] = kernels
    print(json.dumps(out, indent=1))
