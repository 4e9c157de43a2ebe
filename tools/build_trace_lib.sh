#!/bin/bash
# Diagnostic build of the library with -DONMF_TRACE_WG (every workgroup of the product kernels and of
# the dictionary pass records when and where it ran): tools/probe/libonmf_hip_trace.so, read by
# tools/wg_trace.py and tools/sweep_marks.py.  The product library is not touched.
exec "$(dirname "$0")/build_variant_lib.sh" trace -DONMF_TRACE_WG
