#!/bin/bash
# Bench lines of the headline at fixed 10 sweeps, product library and a diagnostic build in turn (three times).
# Usage (GPU box, repo root): bash tools/ab_bench_variant.sh <name> gpurun_out/<dir> [bench arguments]
set -u
NAME=$1; O=${2:-gpurun_out/abv_$NAME}; shift 2; mkdir -p $O
VAR=$PWD/tools/probe/libonmf_hip_$NAME.so
[ -f "$VAR" ] || { echo "missing $VAR"; exit 1; }
: > $O/res.txt
for rep in 1 2 3; do
    for which in base $NAME; do
        if [ $which = base ]; then unset ONMF_HIP_LIB; else export ONMF_HIP_LIB=$VAR; fi
        timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-also --no-parity --stopping-diff 0 "$@" 2>>$O/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline', {}); print('$which', round(d['ms_per_step'], 4), 'sweep launch ms', round(r.get('launch_ms', 0) or 0, 4), 'achieved', round(r.get('achieved', 0), 2))" >> $O/res.txt || exit 1
    done
done
sort $O/res.txt
