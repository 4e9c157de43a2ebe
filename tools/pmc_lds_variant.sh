#!/bin/bash
# The sweep kernel's LDS counters for a diagnostic build (tools/build_variant_lib.sh) or, with name "base", the
# product library.  Usage (GPU box, repo root): bash tools/pmc_lds_variant.sh <name> <r> gpurun_out/<dir>
set -u
NAME=$1; R=$2; O=${3:-gpurun_out/pmc_lds}; mkdir -p $O
export TMPDIR=/tmp
ROOT=$PWD
if [ "$NAME" != base ]; then export ONMF_HIP_LIB=$PWD/tools/probe/libonmf_hip_$NAME.so; [ -f "$ONMF_HIP_LIB" ] || exit 1; fi
( cd /tmp && timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d "$ROOT/$O/pmc_${NAME}_r$R" --output-format csv -- python3 $ROOT/tools/prof_sweep.py --reps 4 --tol 0.01 --r $R ) >> $O/pmc.log 2>&1 || exit 1
python3 - "$O/pmc_${NAME}_r$R" <<'PY'
import csv, glob, collections, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(float)
for row in csv.DictReader(open(f)):
    if "sweep_tile" in row["Kernel_Name"]:
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
print(d, {a: round(b) for a, b in sorted(acc.items())})
if acc.get("SQ_INSTS_LDS"):
    print("   conflict cycles / LDS instruction = %.3f, conflict / LDS array cycles = %.3f, LDS array cycles / SQ busy cycles = %.3f"
          % (acc["SQ_LDS_BANK_CONFLICT"] / acc["SQ_INSTS_LDS"], acc["SQ_LDS_BANK_CONFLICT"] / max(acc["SQ_LDS_IDX_ACTIVE"], 1),
             acc["SQ_LDS_IDX_ACTIVE"] / acc["SQ_BUSY_CYCLES"]))
PY
