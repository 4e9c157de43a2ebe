#!/bin/bash
# Everything profiles/ keeps for a round, in one go (GPU box, repo root):  bash tools/collect_round.sh gpurun_out/<dir>
# Afterwards, here:  python3 tools/refresh_profiles.py gpurun_out/<dir>/bench r03 ; python3 tools/pmc_summary.py
# gpurun_out/<dir>/pmc profiles/r03_pmc ; python3 tools/step_hbm.py gpurun_out/<dir>/step_hbm profiles/r03_step_hbm_traffic.json
set -u
OUT=${1:-gpurun_out/round}
mkdir -p "$OUT"
ROOT=$(pwd)
export TMPDIR=/tmp
if [ "${SKIP_FIRST:-0}" != 1 ]; then     # (SKIP_FIRST=1: only what follows the counter passes)
bash tools/collect_profiles.sh "$OUT/bench" || { echo "bench collection failed"; exit 1; }
bash tools/run_pmc.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1 || { echo "pmc passes failed"; exit 1; }
bash tools/run_step_hbm.sh "$OUT/step_hbm" > "$OUT/step_hbm.log" 2>&1 || { echo "step hbm passes failed"; exit 1; }
fi
# the small reference-scale workload: one step's kernels in time order
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d "$ROOT/$OUT/c2prof" -o bench -- \
    python3 "$ROOT/bench.py" --workload c2 --no-cpu-baseline --no-parity --steps 50 --warmup 10 ) > "$OUT/c2_under_rocprof.json" 2> "$OUT/c2prof.err" || exit 1
python3 tools/step_timeline.py "$OUT/c2prof/bench_results.db" 3 "sweep_tile_kernel<4," > "$OUT/c2_timeline.txt" 2>&1 || exit 1
# Online_NTF step with the LARS coder: kernels of a step
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d "$ROOT/$OUT/ntfprof" -o bench -- \
    python3 "$ROOT/bench.py" --workload c3-ntf --steps 20 --warmup 5 ) > "$OUT/ntf_under_rocprof.json" 2> "$OUT/ntfprof.err" || exit 1
python3 tools/kernel_stats.py "$OUT/ntfprof" > "$OUT/ntf_kernel_stats.txt" 2>&1 || exit 1
# callers that hand over NumPy arrays / want NumPy's random stream
timeout -k 10 300 python3 tools/numpy_caller_time.py > "$OUT/numpy_caller.txt" 2> "$OUT/numpy_caller.err" || exit 1
# a long stream: step time and sweep counts over 100 steps
timeout -k 10 300 python3 tools/long_stream.py > "$OUT/long_stream.txt" 2> "$OUT/long_stream.err" || exit 1
# two ranks sharing the one card (gloo): the multi-rank paths end to end, not a scaling figure
: > "$OUT/gpus_rehearsal.jsonl"
timeout -k 10 400 python3 bench.py --gpus 2 --workload c4 --steps 8 --warmup 2 >> "$OUT/gpus_rehearsal.jsonl" 2> "$OUT/rehearsal.err" || exit 1
timeout -k 10 400 python3 bench.py --gpus 2 --workload c4 --mode sharded --steps 8 --warmup 2 >> "$OUT/gpus_rehearsal.jsonl" 2>> "$OUT/rehearsal.err" || exit 1
timeout -k 10 300 python3 bench.py --gpus 2 --workload c5 --steps 20 --warmup 5 >> "$OUT/gpus_rehearsal.jsonl" 2>> "$OUT/rehearsal.err" || exit 1
echo round collected
