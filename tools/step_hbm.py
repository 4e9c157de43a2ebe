#!/usr/bin/env python3
"""HBM bytes per step of the default bench workload over ALL kernels of a step, from two rocprofv3 passes of
bench.py itself (FETCH_SIZE in one run, WRITE_SIZE in another; tools/run_step_hbm.sh):

    python3 tools/step_hbm.py gpurun_out/<dir> profiles/r03_step_hbm_traffic.json

FETCH_SIZE is in KiB and reports half of the bytes on gfx950 (doubled here, as for the sweep kernel's own
record); WRITE_SIZE is in KiB as read.  Kernels of the library and the runtime's buffer copies are counted;
torch's kernels (the synthetic batch, drawn once before the steps) are not.  Each counter is read from a short and
a long run of the same command; the difference divided by the difference in steps is what one step moves (the
set-up work of a run - first dictionary, probes, warm-up of the stopping test - drops out).
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def totals(path, counter):
    by = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                n = r["Kernel_Name"]
                if "at::native" in n or "at::cuda" in n or n.startswith("void at::"):
                    continue
                key = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                c = by.setdefault(key, [0.0, 0])
                c[0] += float(r["Counter_Value"])
                c[1] += 1
    return by


def main():
    src, out = sys.argv[1], sys.argv[2]
    meta = json.load(open(os.path.join(src, "run.json")))
    steps = meta["steps_long"] - meta["steps_short"]
    d, r, b = bench.WORKLOADS["headline"]

    def diff(name, counter):
        lo, hi = totals(os.path.join(src, name + "_short"), counter), totals(os.path.join(src, name + "_long"), counter)
        return {k: [hi[k][0] - lo.get(k, [0.0, 0])[0], hi[k][1] - lo.get(k, [0.0, 0])[1]] for k in hi}
    fetch, write = diff("f", "FETCH_SIZE"), diff("w", "WRITE_SIZE")
    per_kernel = {}
    for k in sorted(set(fetch) | set(write)):
        rd = 2048.0 * fetch.get(k, [0.0, 0])[0] / steps
        wr = 1024.0 * write.get(k, [0.0, 0])[0] / steps
        per_kernel[k] = {"read_bytes_per_step": rd, "write_bytes_per_step": wr,
                         "launches_per_step": max(fetch.get(k, [0, 0])[1], write.get(k, [0, 0])[1]) / steps}
    total = sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in per_kernel.values())
    sweeps = meta.get("mean_sweeps", 10.0)
    # what a step must move at least: X read once per product that needs it fused into one pass (d*b), H0 never
    # materialised, codes written once (r*b), the aggregates and the dictionary (small)
    algorithmic = 4.0 * (d * b + r * b + 2 * (r * r + r * d) + d * r)
    rec = {"command": meta["command"], "shape": {"d": d, "r": r, "b": b}, "steps_counted": steps, "mean_sweeps": sweeps,
           "hbm_bytes_per_step": total, "algorithmic_bytes_per_step": algorithmic,
           "ratio_to_algorithmic": total / algorithmic, "per_kernel": per_kernel, "csrc_sha256": bench.csrc_sha256(),
           "note": "FETCH_SIZE (KiB) doubled (gfx950 correction), WRITE_SIZE (KiB) as read; counters from two separate "
                   "rocprofv3 --pmc runs of the same command; kernels run one at a time under the counters, so the bytes "
                   "are those of the kernels alone (overlap of the dictionary pass with the products does not change them)"}
    json.dump(rec, open(out, "w"), indent=1)
    print("bytes per step %.3f GB (%.1fx the %.3f GB a fused step must move)" % (total / 1e9, total / algorithmic, algorithmic / 1e9))
    for k, v in sorted(per_kernel.items(), key=lambda kv: -(kv[1]["read_bytes_per_step"] + kv[1]["write_bytes_per_step"]))[:12]:
        print("  %-60s read %8.1f MB  write %8.1f MB  launches/step %.2f" % (k[:60], v["read_bytes_per_step"] / 1e6, v["write_bytes_per_step"] / 1e6, v["launches_per_step"]))


if __name__ == "__main__":
    main()
