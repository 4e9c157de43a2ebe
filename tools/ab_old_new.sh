#!/bin/bash
# Same-box comparison of the library as built against an older build kept as tools/probe/libonmf_hip_old.so:
# hashes of the codes (tools/hash_solve.py) and the bench line at fixed 10 sweeps, new - old - new - old.
set -u
O=${1:-gpurun_out/ab}; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-also --no-parity --stopping-diff 0"
line() { timeout -k 10 200 $B 2>>$O/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['achieved'])" >> $O/res.txt || exit 1; }
: > $O/res.txt
cp onmf_ontf_ndl_amd/libonmf_hip.so /tmp/new.so
timeout -k 10 200 python3 tools/hash_solve.py > $O/hash_new.txt 2>>$O/err.txt || exit 1
line new; 
cp tools/probe/libonmf_hip_old.so onmf_ontf_ndl_amd/libonmf_hip.so
timeout -k 10 200 python3 tools/hash_solve.py > $O/hash_old.txt 2>>$O/err.txt || exit 1
line old
cp /tmp/new.so onmf_ontf_ndl_amd/libonmf_hip.so; line new
cp tools/probe/libonmf_hip_old.so onmf_ontf_ndl_amd/libonmf_hip.so; line old
cp /tmp/new.so onmf_ontf_ndl_amd/libonmf_hip.so; line new
cmp $O/hash_new.txt $O/hash_old.txt && echo "hashes equal" >> $O/res.txt || echo "HASHES DIFFER" >> $O/res.txt
cat $O/res.txt
