#!/bin/bash
# Diagnostic build of the library with -DONMF_TRACE_WG (every workgroup of the product kernels and of
# the dictionary pass records when and where it ran): tools/probe/libonmf_hip_trace.so, read by
# tools/wg_trace.py.  The product library is not touched.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/onmf_ontf_ndl_amd/csrc
OBJ=${TMPDIR:-/tmp}/onmf_trace_objs
mkdir -p "$OBJ" "$ROOT/tools/probe"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -DONMF_TRACE_WG -Wno-unknown-warning-option"
pids=()
for f in api gemm sweep sweep_ct1 sweep_ct2 sweep_ct2s sweep_ct2m sweep_each sweep_each4 dict misc screen patches ndl; do
    ( cd "$SRC" && /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o "$OBJ/$f.o" 2>/dev/null ) &
    pids+=($!)
    if [ ${#pids[@]} -ge 7 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
g++ -O3 -std=c++17 -fPIC -c "$SRC/host_eig.cpp" -o "$OBJ/host_eig.o"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/tools/probe/libonmf_hip_trace.so" "$OBJ"/*.o
echo built "$ROOT/tools/probe/libonmf_hip_trace.so"
