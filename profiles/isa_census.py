"""Instruction census of a kernel from its ISA (no GPU needed): compiles one source file for gfx950 with -DSFM_CENSUS -S (the SFM_PHASE
markers of device_math.hpp become comments between scheduling barriers), cuts the named kernel out of the assembly and counts the
instructions per phase and class.

    python profiles/isa_census.py cuda-sfm_amd/csrc/ransac.hip ransac_solve_lanes1_qr [extra hipcc flags...]

Static counts of the instruction TEXT: a loop body counts once, both sides of a branch count.  The kernels censused here are
straight-line per phase (fully unrolled solvers) or have one hot loop whose body is a phase of its own."""
import collections
import re
import subprocess
import sys
import tempfile
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-DOCML_BASIC_ROUNDED_OPERATIONS",
         "-fPIC", "-fvisibility=hidden", "-Wno-unused-function", "-Wno-pass-failed", "--cuda-device-only", "-S", "-DSFM_CENSUS=1", "-I" + os.path.join(ROOT, "include")]


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk_"): return "valu_packed"
    if re.match(r"v_(fma|mul|add|sub|mac|fmac|mad)_f64|v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|cvt_f64|cvt_f32_f64|trig_preop|ldexp|frexp)_f64|v_.*_f64", op): return "valu_f64"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op): return "valu_trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait_nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"): return "vmem"
    return "other"


def main():
    src, kernel = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", out, os.path.join(ROOT, src)], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    start = next(i for i, ln in enumerate(text) if re.match(r"^_Z\w*%s\w*:" % re.escape(kernel), ln))
    end = next(i for i in range(start, len(text)) if text[i].strip().startswith("s_endpgm"))
    phase = "prologue"
    counts = collections.OrderedDict()
    for ln in text[start + 1:end + 1]:
        t = ln.strip()
        m = re.match(r";\s*##PHASE (\S+)", t)
        if m:
            phase = m.group(1)
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        counts.setdefault(phase, collections.Counter())[classify(op)] += 1
    classes = ["valu", "valu_packed", "valu_f64", "valu_trans", "mfma", "salu", "branch", "wait_nop", "lds", "vmem", "other"]
    meta = [ln.strip() for ln in text if re.search(r"\.(vgpr_count|sgpr_count|private_segment_fixed_size):", ln)]
    print(f"# {kernel} ({src}{' ' + ' '.join(extra) if extra else ''}): static instruction counts per phase")
    print("%-16s" % "phase" + "".join("%12s" % c for c in classes) + "%12s" % "all_valu")
    tot = collections.Counter()
    for ph, c in counts.items():
        allv = c["valu"] + c["valu_packed"] + c["valu_f64"] + c["valu_trans"] + c["mfma"]
        print("%-16s" % ph + "".join("%12d" % c[k] for k in classes) + "%12d" % allv)
        tot.update(c)
    allv = tot["valu"] + tot["valu_packed"] + tot["valu_f64"] + tot["valu_trans"] + tot["mfma"]
    print("%-16s" % "TOTAL" + "".join("%12d" % tot[k] for k in classes) + "%12d" % allv)


if __name__ == "__main__":
    main()
