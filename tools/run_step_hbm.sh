#!/bin/bash
# Two counter passes (FETCH_SIZE, WRITE_SIZE; one per run, the program itself right after `--`) over the default
# bench workload, for tools/step_hbm.py.  Usage (GPU box, repo root): bash tools/run_step_hbm.sh gpurun_out/<dir>
set -u
OUT=${1:-gpurun_out/step_hbm}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
STEPS=6; WARM=2
ARGS="--no-cpu-baseline --no-also --no-parity --steps $STEPS --warmup $WARM"
( cd /tmp && timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d "$ROOT/$OUT/fetch" --output-format csv -- python3 $ROOT/bench.py $ARGS ) > "$OUT/fetch_line.json" 2> "$OUT/fetch.err" || exit 1
( cd /tmp && timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d "$ROOT/$OUT/write" --output-format csv -- python3 $ROOT/bench.py $ARGS ) > "$OUT/write_line.json" 2> "$OUT/write.err" || exit 1
python3 - "$OUT" "$STEPS" "$WARM" "$ARGS" <<'PY'
import json, sys
out, steps, warm, args = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
line = json.loads(open(out + "/fetch_line.json").read().strip().splitlines()[-1])
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py " + args, "steps": steps, "warmup": warm,
           "mean_sweeps": line["config"]["mean_sweeps"]}, open(out + "/run.json", "w"))
PY
echo passes done
