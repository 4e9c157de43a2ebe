#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes that tools/run_pmc.sh left under a directory into one JSON per
kernel family: per counter the mean over that kernel's dispatches, plus the derived figures the
roofline discussion uses.  Usage: python3 tools/pmc_summary.py gpurun_out/<dir> profiles/r02_pmc"""
import csv
import glob
import hashlib
import json
import os
import sys

CU = 256          # compute units of an MI355X
PICK = {          # family -> (substring of the kernel name, passes that ran it)
    "sweep": ("sweep_tile_kernel", ["sweep_sq1", "sweep_sq2", "sweep_fetch", "sweep_write"]),
    "gemm_tt": ("gemm_f32_kernel<true, true,", ["gemm_sq1", "gemm_sq2", "gemm_fetch", "gemm_write"]),
    "gemm_ff": ("gemm_f32_kernel<false, false,", ["gemm_sq1", "gemm_sq2", "gemm_fetch", "gemm_write"]),
}


def read_pass(path):
    rows = []
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            rows += list(csv.DictReader(fh))
    return rows


def main():
    src, out = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for fam, (needle, passes) in PICK.items():
        rec = {"kernel_name_contains": needle, "counters": {}, "dispatches": {}}
        for p in passes:
            rows = [r for r in read_pass(os.path.join(src, p)) if needle in r["Kernel_Name"]]
            if not rows:
                continue
            # the longest-running variant of the family (largest grid x most dispatches) is the one of interest
            by = {}
            for r in rows:
                by.setdefault(r["Counter_Name"], {}).setdefault(r["Kernel_Name"], {}).setdefault(
                    r["Dispatch_Id"], 0.0)
                by[r["Counter_Name"]][r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            for cname, kernels in by.items():
                kname = max(kernels, key=lambda k: sum(kernels[k].values()))
                vals = list(kernels[kname].values())
                rec.setdefault("all_dispatches", {})[cname] = vals
                # the full-length launches only (a solve that refreshes its probes cuts its launch short)
                vals = [v for v in vals if v >= 0.8 * max(vals)]
                rec["counters"][cname] = sum(vals) / len(vals)
                rec["dispatches"][cname] = len(vals)
                rec["kernel"] = kname
        c = rec["counters"]
        d = {}
        if "SQ_INSTS_MFMA" in c and "SQ_INSTS_VALU" in c:
            d["valu_instructions_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"] \
                if c["SQ_INSTS_VALU"] > c["SQ_INSTS_MFMA"] else c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]
            d["note_valu"] = "SQ_INSTS_VALU counts MFMA instructions too when it exceeds SQ_INSTS_MFMA; the ratio is of the others"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # rocprofv3's MfmaUtil: sum of busy cycles over the SIMDs / (GRBM_GUI_ACTIVE of ONE XCD x SIMDs);
            # the CSV holds GRBM_GUI_ACTIVE summed over the 8 XCDs
            d["mfma_busy_pct"] = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * CU * 4)
            d["note_busy"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)"
        if "SQ_LDS_BANK_CONFLICT" in c and "SQ_INSTS_LDS" in c and c["SQ_INSTS_LDS"]:
            d["lds_bank_conflict_cycles_per_lds_instruction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"]
        if "FETCH_SIZE" in c:
            d["hbm_read_bytes_per_dispatch_corrected"] = 2.0 * 1024.0 * c["FETCH_SIZE"]
            d["note_fetch"] = "FETCH_SIZE is in KiB and reports half of the bytes on gfx950: doubled"
        if "WRITE_SIZE" in c:
            d["hbm_write_bytes_per_dispatch"] = 1024.0 * c["WRITE_SIZE"]
        if fam == "sweep" and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            # launches of 2 sweeps (probe refresh) and of 10: per-sweep bytes from their difference
            f_all, w_all = rec["all_dispatches"]["FETCH_SIZE"], rec["all_dispatches"]["WRITE_SIZE"]
            short = lambda v: [x for x in v if x < 0.5 * max(v)]
            full = lambda v: [x for x in v if x >= 0.8 * max(v)]
            if short(f_all) and short(w_all):
                mean = lambda v: sum(v) / len(v)
                b2 = 2048.0 * mean(short(f_all)) + 1024.0 * mean(short(w_all))
                b10 = 2048.0 * mean(full(f_all)) + 1024.0 * mean(full(w_all))
                d["hbm_bytes_2_sweep_launch"] = b2
                d["hbm_bytes_10_sweep_launch"] = b10
                d["hbm_bytes_per_sweep_in_launch"] = (b10 - b2) / 8.0
                d["hbm_bytes_first_read"] = b2 - 2.0 * (b10 - b2) / 8.0
        rec["derived"] = d
        srcs = {"sweep": "sweep_tile.h", "gemm_tt": "gemm.hip", "gemm_ff": "gemm.hip"}[fam]
        with open(os.path.join(root, "onmf_ontf_ndl_amd", "csrc", srcs), "rb") as fh:
            rec["kernel_source"] = srcs
            rec["kernel_source_sha256"] = hashlib.sha256(fh.read()).hexdigest()
        with open("%s_%s.json" % (out, fam), "w") as fh:
            json.dump(rec, fh, indent=1)
        if fam == "sweep" and "hbm_bytes_per_sweep_in_launch" in d:
            # the record bench.py reads for roofline.traffic (it checks the hash of the kernel's source)
            tr = {"kernel": rec.get("kernel"), "shape": {"d": 1024, "r": 256, "b": 65536},
                  "command": "bash tools/run_pmc.sh (rocprofv3 --pmc FETCH_SIZE -- python3 tools/prof_sweep.py --reps 4 "
                             "--tol 0.01; WRITE_SIZE in a second run)"}
            for k in ("hbm_bytes_2_sweep_launch", "hbm_bytes_10_sweep_launch", "hbm_bytes_per_sweep_in_launch",
                      "hbm_bytes_first_read"):
                tr[k] = d[k]
            tr["algorithmic_bytes"] = {"first_read": 4 * 256 * 65536, "per_sweep": 2 * 4 * 256 * 65536}
            tr["kernel_source"], tr["kernel_source_sha256"] = rec["kernel_source"], rec["kernel_source_sha256"]
            tr["note"] = ("FETCH_SIZE (KiB) doubled as the gfx950 correction prescribes (the kernel's loads are 4 B per "
                          "lane, a width the correction was not calibrated on); WRITE_SIZE (KiB) as read. Per-sweep bytes "
                          "from the difference of the 10-sweep launches and the 2-sweep launches (solves that refresh "
                          "their probes cut the launch after the second sweep). The first read of H comes out below its "
                          "67 MB: the start code was written just before by the fill kernel and part of it is still on "
                          "the die.")
            with open(out.rsplit("_", 1)[0] + "_sweep_hbm_traffic.json", "w") as fh:
                json.dump(tr, fh, indent=1)
        print(fam, json.dumps(rec["derived"]))


if __name__ == "__main__":
    main()
