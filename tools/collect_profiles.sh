#!/bin/bash
# One collection of what profiles/ keeps for a round: the default bench line (alone and under
# rocprofv3 --kernel-trace --stats), one step's kernel timeline, the other workloads' lines.
# Usage (on the GPU box, from the repo root):  bash tools/collect_profiles.sh gpurun_out/<dir>
# then here:  python3 tools/refresh_profiles.py gpurun_out/<dir> r02
set -u
OUT=${1:-gpurun_out/collect}
mkdir -p "$OUT"
ROOT=$(pwd)
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err" || exit 1
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/prof" -o bench -- \
    python3 "$ROOT/bench.py" --no-cpu-baseline --no-also --no-parity ) > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err" || exit 1
python3 tools/step_timeline.py "$OUT/prof/bench_results.db" 3 > "$OUT/timeline.txt" 2>&1 || exit 1
: > "$OUT/workloads.txt"
for w in c1 c2 c3 c4 c5; do
    timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --no-parity >> "$OUT/workloads.txt" 2>> "$OUT/workloads.err" || exit 1
done
timeout -k 10 300 python3 bench.py --stopping-diff 0 --no-cpu-baseline --no-also --no-parity >> "$OUT/workloads.txt" 2>> "$OUT/workloads.err" || exit 1
for w in c3-ntf-100 c3-ntf; do
    timeout -k 10 400 python3 bench.py --workload $w >> "$OUT/workloads.txt" 2>> "$OUT/workloads.err" || exit 1
done
echo collected
