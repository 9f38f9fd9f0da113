"""DESIGN.md section 5's table, generated from the committed evidence so that the numbers cannot drift from the files.

    python profiles/make_design_table.py            # print the table
    python profiles/make_design_table.py --write    # replace the block between <!-- r06-table:begin --> and <!-- r06-table:end --> in DESIGN.md

Sources (all under profiles/, written by profiles/collect_r06.sh on ONE MI355X box): r06_bench_line.json (the default command: headline,
cpu_baseline, extra configurations, projection), r06_bench_line_serial.json, r06_bench_rank8.json / _serial, r06_bench_c3.json, r06_bench_c4.json,
r06_bench_c4_rank8.json, r06_bench_*_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same commands), r06_traffic.json (--pmc passes)."""
import csv
import json
import os
import sys

P = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(P)


def line(name):
    with open(os.path.join(P, name)) as f:
        rows = [ln for ln in f.read().strip().splitlines() if ln.startswith("{")]
    return json.loads(rows[-1])


def kernel_avg_us(csv_name, needle):
    with open(os.path.join(P, csv_name)) as f:
        for r in csv.DictReader(f):
            if needle in r["Name"]:
                return float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, int(r["Calls"])
    return None, None, 0


def fmt(x, nd=4):
    return "n/a" if x is None else f"{x:.{nd}f}"


def main():
    head = line("r06_bench_line.json")
    ser = line("r06_bench_line_serial.json")
    r8 = line("r06_bench_rank8.json")
    r8s = line("r06_bench_rank8_serial.json")
    c3 = line("r06_bench_c3.json")
    c4 = line("r06_bench_c4.json")
    c4r = line("r06_bench_c4_rank8.json")
    traffic = json.load(open(os.path.join(P, "r06_traffic.json")))
    rows = []

    def row(label, d, stats_csv, src, overlapped_csv=None):
        r = d["roofline"]
        avg, mn, calls = kernel_avg_us(stats_csv, "ransac_score_prefilter") if stats_csv else (None, None, 0)
        prof = f"{avg:.1f} us avg / {mn:.1f} min over {calls} launches (`{stats_csv}`)" if avg else "-"
        if overlapped_csv:          # pipelined steps: the solve of the next step runs beside the scoring launch, whose profiled duration grows with it
            oavg, omn, ocalls = kernel_avg_us(overlapped_csv, "ransac_score_prefilter")
            prof += f"; with the next step's solve beside it {oavg:.1f} us avg / {omn:.1f} min over {ocalls} (`{overlapped_csv}`)"
        rows.append(f"| {label} | {fmt(d['ms_per_step'])} | {d['value']:.3g} | {fmt(r['avg_launch_ms'])} ms live at {r['shader_clock_mhz']:.0f} MHz; rocprofv3: {prof} | "
                    f"{fmt(r['solve_kernel_avg_ms'])} | {r['frac']:.3f} ({fmt(r.get('frac_at_sustained_clock'), 3)} at the sustained clock) | `{src}` |")

    row("headline 4096 x 2^20, pipelined (the contractual line; the launch is timed live in serial steps of the same run)", head, "r06_bench_headline_kernel_stats.csv", "r06_bench_line.json",
        overlapped_csv="r06_bench_pipelined_kernel_stats.csv")
    row("headline, `--serial`", ser, "r06_bench_headline_kernel_stats.csv", "r06_bench_line_serial.json")
    row("one of 8 ranks' share (131072 hypotheses), pipelined", r8, "r06_bench_rank8_kernel_stats.csv", "r06_bench_rank8.json")
    row("the same, `--serial`", r8s, None, "r06_bench_rank8_serial.json")
    row("c3 16384 x 65536", c3, "r06_bench_c3_kernel_stats.csv", "r06_bench_c3.json")
    row("c4 16384 x 2^20 on one GPU", c4, "r06_bench_c4_kernel_stats.csv", "r06_bench_c4.json")
    row("c4, one of 8 ranks' share", c4r, None, "r06_bench_c4_rank8.json")
    out = ["| configuration (1 x MI355X, one box) | ms / step | hypotheses / s | scoring launch | solve launch, ms | roofline frac (fp16 MFMA floor / launch) | file |",
           "|---|---|---|---|---|---|---|"] + rows
    rl = head["roofline"]
    pf = traffic.get("ransac_score_prefilter", {})
    sv = traffic.get("ransac_solve_lanes1_qr", {})
    out.append("")
    out.append(f"Roofline of the dominant kernel on the contractual line: bound `{rl['bound']}`, floor {rl['floor_ms']:.4f} ms (scan floor {rl['valu_scan_floor_ms']:.4f} ms), "
               f"achieved {rl['achieved']:.0f} of {rl['peak']:.0f} TFLOP/s fp16 MFMA = **{rl['frac']:.3f}**; SURVEY 8(d) ratio {rl['survey_8d_ratio']:.2f} (> 1: flagged, not a fraction). "
               f"Counters (`r06_traffic.json`, sources {traffic.get('code_sha256_16')}): vector issue {pf.get('valu_busy_frac')} busy, matrix pipe {pf.get('mfma_busy_frac')}, LDS {pf.get('lds_busy_frac')} "
               f"(bank conflicts {pf.get('lds_bank_conflict_frac')} of its cycles), {pf.get('valu_insts_per_launch', 0) / 1e8:.3f}e8 vector instructions per launch; a wavefront issues "
               f"{pf.get('wave_issuing_frac')} of its cycles, is parked at s_waitcnt {pf.get('wave_parked_at_waitcnt_frac')}, stalled at issue {pf.get('wave_issue_stalled_frac')}. "
               f"HBM traffic per step {rl.get('traffic_step_bytes', 0) / 1e6:.0f} MB = {rl.get('traffic_step_over_algorithmic', 0):.2f} x algorithmic "
               f"(scoring {(pf.get('fetch_kb', 0) + pf.get('write_kb', 0)) / 1e3:.0f} MB, solve {(sv.get('fetch_kb', 0) + sv.get('write_kb', 0)) / 1e3:.0f} MB: {sv.get('write_kb', 0) * 1024 / 2 ** 20:.0f} B written per hypothesis).")
    cb = head["cpu_baseline"]
    sp = head.get("scaling_projection", {})
    out.append(f"CPU port in the same run: {cb['value']:.3g} hypotheses/s on {cb['cores']} threads ({cb['kind']}; OpenCV: {cb.get('opencv_findEssentialMat')}). "
               f"Exchange step alone (one-rank communicator): {head.get('exchange_us'):.1f} us. Projection to 8 GPUs from this box (UNMEASURED ON HARDWARE): "
               f"{sp.get('pipelined', {}).get('excl_exchange', 0):.2f}x before the exchange, {sp.get('pipelined', {}).get('incl_exchange_as_measured_with_one_rank', 0):.2f}x with it as measured, "
               f"{sp.get('pipelined', {}).get('incl_exchange_assumed_20us', 0):.2f}x with an assumed 20 us; configs[3]: "
               f"{sp.get('configs3_16384_matches', {}).get('pipelined', {}).get('excl_exchange', 0):.2f}x / {sp.get('configs3_16384_matches', {}).get('pipelined', {}).get('incl_exchange_assumed_20us', 0):.2f}x.")
    ex = head.get("extra", {})
    m = ex.get("match_2048", {}); d1 = ex.get("c1_dino_pair", {}).get("ms", {})
    out.append(f"Neighbouring rows in the same line: 2048^2 match {1e3 * m.get('ms', 0):.1f} us, 4096^2 {1e3 * ex.get('match_4096', {}).get('ms', 0):.1f} us, 16384^2 {1e3 * ex.get('match_16384', {}).get('ms', 0):.1f} us; "
               f"the dino pair end to end {1e3 * d1.get('match_fillXU_estimateE_pose_chain', 0):.1f} us (match {1e3 * d1.get('match', 0):.1f}, fillXU {1e3 * d1.get('fillXU', 0):.1f}, estimateE {1e3 * d1.get('estimateE', 0):.1f}, pose chain {1e3 * d1.get('pose_chain', 0):.1f}); "
               f"every `extra` entry carries `parity_vs_oracle: true` from a full sweep.")
    text = "\n".join(out)
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        doc = open(path).read()
        b, e = "<!-- r06-table:begin -->", "<!-- r06-table:end -->"
        i, j = doc.index(b) + len(b), doc.index(e)
        open(path, "w").write(doc[:i] + "\n" + text + "\n" + doc[j:])
        print("DESIGN.md updated")
    else:
        print(text)


if __name__ == "__main__":
    main()
