#!/bin/bash
# Counter passes (FETCH_SIZE, WRITE_SIZE; one per run, the program itself right after `--`) over the default bench
# workload at two lengths (what a step moves is the difference: set-up work drops out), for tools/step_hbm.py.  Usage (GPU box, repo root): bash tools/run_step_hbm.sh gpurun_out/<dir>
set -u
OUT=${1:-gpurun_out/step_hbm}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
SHORT=4; LONG=12; WARM=2
ARGS="--no-cpu-baseline --no-also --no-parity --warmup $WARM"
for c in FETCH_SIZE WRITE_SIZE; do
    n=$(echo $c | cut -c1 | tr FW fw)
    ( cd /tmp && timeout -s KILL 300 rocprofv3 --pmc $c -d "$ROOT/$OUT/${n}_short" --output-format csv -- python3 $ROOT/bench.py $ARGS --steps $SHORT ) > "$OUT/${n}_short_line.json" 2> "$OUT/${n}_short.err" || exit 1
    ( cd /tmp && timeout -s KILL 300 rocprofv3 --pmc $c -d "$ROOT/$OUT/${n}_long" --output-format csv -- python3 $ROOT/bench.py $ARGS --steps $LONG ) > "$OUT/${n}_long_line.json" 2> "$OUT/${n}_long.err" || exit 1
done
python3 - "$OUT" "$SHORT" "$LONG" "$ARGS" <<'PY'
import json, sys
out, short, long_, args = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
line = json.loads(open(out + "/f_long_line.json").read().strip().splitlines()[-1])
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py %s --steps %d | %d" % (args, short, long_),
           "steps_short": short, "steps_long": long_, "mean_sweeps": line["config"]["mean_sweeps"]}, open(out + "/run.json", "w"))
PY
echo passes done
