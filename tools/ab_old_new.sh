#!/bin/bash
# Same-box comparison of the library as built against an older build kept as tools/probe/libonmf_hip_old.so:
# hashes of the codes (tools/hash_solve.py) and bench lines (the headline at fixed 10 sweeps and with the
# reference's stopping test, and the small workloads), new and old in turn.
set -u
O=${1:-gpurun_out/ab}; mkdir -p $O
line() {   # label, bench arguments...
    local label=$1; shift
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-also --no-parity "$@" 2>>$O/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', '$*', round(d['ms_per_step'], 4), round(d.get('roofline', {}).get('achieved', 0), 2))" >> $O/res.txt || exit 1
}
: > $O/res.txt
# the build to measure is chosen through ONMF_HIP_LIB (read by onmf_ontf_ndl_amd/_lib.py); the product library
# in the package directory is never overwritten
use() { if [ "$1" = new ]; then unset ONMF_HIP_LIB; else export ONMF_HIP_LIB=$PWD/tools/probe/libonmf_hip_old.so; fi; }
use new; timeout -k 10 200 python3 tools/hash_solve.py > $O/hash_new.txt 2>>$O/err.txt || exit 1
use old; timeout -k 10 200 python3 tools/hash_solve.py > $O/hash_old.txt 2>>$O/err.txt || exit 1
for rep in 1 2 3; do
    for which in new old; do
        use $which
        line $which --stopping-diff 0
        line $which
        for w in c1 c2 c3 c5; do line $which --workload $w --steps 200 --warmup 20; done
    done
done
use new
cmp $O/hash_new.txt $O/hash_old.txt && echo "hashes equal" >> $O/res.txt || echo "HASHES DIFFER" >> $O/res.txt
sort -k2,3 -k1,1 $O/res.txt
