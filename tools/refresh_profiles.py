#!/usr/bin/env python3
"""Turn one collection of bench outputs (a directory with bench_line.json, bench_under_rocprof.json,
workloads.txt, timeline.txt and prof/bench_results.db from `rocprofv3 --kernel-trace --stats`) into the
files kept under profiles/.  Usage: python3 tools/refresh_profiles.py gpurun_out/<dir> [prefix]"""
import csv
import json
import os
import shutil
import sqlite3
import statistics
import sys


def main():
    src = sys.argv[1].rstrip("/") + "/"
    prefix = sys.argv[2] if len(sys.argv) > 2 else "r01"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles", prefix + "_")
    db = sqlite3.connect(src + "prof/bench_results.db")
    rows = list(db.execute("select name, start, end from kernels order by start"))
    by = {}
    for n, s, e in rows:
        by.setdefault(n, []).append(e - s)
    tot = sum(sum(v) for v in by.values())
    with open(out + "bench_default_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 3), round(100 * sum(v) / tot, 2), min(v), max(v),
                        round(statistics.pstdev(v), 3)])
    u = json.loads(open(src + "bench_under_rocprof.json").read().strip().splitlines()[-1])
    db.execute("select 1")
    v = [e - s for n, s, e in rows if "sweep_tile_kernel<16, true, true, 2" in n]
    kname = sorted({n for n, s, e in rows if "sweep_tile_kernel<16, true, true, 2" in n})[0]
    n = u["roofline"]["launches"]
    json.dump({"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline (run from /tmp)",
               "kernel": kname, "launch_ns_in_order": v,
               "note": "the last %d launches belong to the %d timed steps, the ones before to the warm-up steps"
                       % (n, u["steps"]),
               "timed_region_launches": n, "timed_region_avg_ms_from_trace": sum(v[-n:]) / n / 1e6,
               "timed_region_avg_ms_live_hip_events": u["roofline"]["avg_launch_ms"]},
              open(out + "bench_default_sweep_launches.json", "w"), indent=1)
    shutil.copy(src + "bench_line.json", out + "bench_default_line.json")
    shutil.copy(src + "bench_under_rocprof.json", out + "bench_under_rocprof_line.json")
    shutil.copy(src + "timeline.txt", out + "bench_step_timeline.txt")
    lines = [json.loads(l) for l in open(src + "workloads.txt") if l.strip().startswith("{")]
    json.dump(lines, open(out + "bench_other_workloads.json", "w"))
    for l in lines:
        print(l["config"]["workload"][:12], round(l["ms_per_step"], 3), round(l["value"] / 1e6, 3),
              round(l["config"].get("step_tflops_per_gpu", 0), 2),
              round(l["config"].get("step_frac_of_fp32_mfma_peak", 0), 3))
    d = json.loads(open(src + "bench_line.json").read().strip().splitlines()[-1])
    print("default:", d["ms_per_step"], d["value"], d["config"]["step_tflops_per_gpu"],
          d["config"]["step_frac_of_fp32_mfma_peak"], d["roofline"], d.get("also"), d.get("cpu_baseline"))
    print("under rocprof:", u["ms_per_step"], "trace avg launch ms", sum(v[-n:]) / n / 1e6)


if __name__ == "__main__":
    main()
