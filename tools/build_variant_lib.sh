#!/bin/bash
# Diagnostic build of the library with extra compiler flags: tools/probe/libonmf_hip_<name>.so, chosen at run
# time through ONMF_HIP_LIB.  The product library is not touched.
# Usage: tools/build_variant_lib.sh <name> [flags...]      e.g.  tools/build_variant_lib.sh rowswap -DONMF_GS_ROWSWAP=1
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/onmf_ontf_ndl_amd/csrc
OBJ=${TMPDIR:-/tmp}/onmf_variant_objs_$NAME
mkdir -p "$OBJ" "$ROOT/tools/probe"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unknown-warning-option $*"
pids=()
for f in $(cd "$SRC" && ls *.hip); do
    ( cd "$SRC" && /opt/rocm/bin/hipcc $FLAGS -c $f -o "$OBJ/${f%.hip}.o" 2>/dev/null ) &
    pids+=($!)
    if [ ${#pids[@]} -ge 7 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
g++ -O3 -std=c++17 -fPIC -c "$SRC/host_eig.cpp" -o "$OBJ/host_eig.o"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/tools/probe/libonmf_hip_$NAME.so" "$OBJ"/*.o
echo built "$ROOT/tools/probe/libonmf_hip_$NAME.so"
