#!/bin/bash
# Counter passes (rocprofv3 --pmc, one pass per run, the program itself right after `--`) over the three
# dominant kernels: the code sweeps (tools/prof_sweep.py) and the two product kernels (tools/gemm_pmc.py).
# Usage (on the GPU box, from the repo root):  bash tools/run_pmc.sh gpurun_out/<dir>
# Every pass runs under its own time limit; the script stops at the first pass that fails.
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
pass() {   # name, counters, program...
    local name=$1 counters=$2
    shift 2
    echo "== pass $name: $counters" | tee -a "$OUT/passes.log"
    ( cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $counters -d "$ROOT/$OUT/$name" --output-format csv -- "$@" ) \
        >> "$OUT/passes.log" 2>&1
    local rc=$?
    echo "rc=$rc" | tee -a "$OUT/passes.log"
    return $rc
}
SWEEP="python3 $ROOT/tools/prof_sweep.py --reps 4 --tol 0.01"
GEMM="python3 $ROOT/tools/gemm_pmc.py"
pass sweep_sq1 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" $SWEEP &&
pass sweep_sq2 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" $SWEEP &&
pass sweep_fetch "FETCH_SIZE" $SWEEP &&
pass sweep_write "WRITE_SIZE" $SWEEP &&
pass gemm_sq1 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" $GEMM &&
pass gemm_sq2 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" $GEMM &&
pass gemm_fetch "FETCH_SIZE" $GEMM &&
pass gemm_write "WRITE_SIZE" $GEMM
echo "done rc=$?" | tee -a "$OUT/passes.log"
