#!/usr/bin/env python3
"""Where a sweep of the one-launch solve spends its time: workgroup 0's time marks (diagnostic build,
tools/build_trace_lib.sh) - sweep start, after the start barrier, after the block loop, after the
workgroup's partial sums, after the barrier behind them, after the look at the tested sweep's count, after the
test evaluated ahead, after the atomics' return, after the last barrier."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from onmf_ontf_ndl_amd import _lib

_lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "libonmf_hip_trace.so")
from onmf_ontf_ndl_amd.engine import get_engine

d, r, b = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 64, 8192)
eng = get_engine()
lib = C.CDLL(_lib.LIB_PATH)
lib.onmf_debug_sweep_marks.restype = C.c_int
lib.onmf_debug_sweep_marks.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
g = torch.Generator(device=eng.device).manual_seed(1)
X = torch.rand(d, b, device=eng.device, generator=g)
W = eng.update_dict(torch.rand(d, r, device=eng.device, generator=g), eng.zeros(r, r), eng.zeros(r, d))
names = ["start barrier", "block loop", "partial sums", "barrier", "atomics out, count of the tested sweep in",
         "test ahead", "atomics back", "barrier"]
ORDER = [0, 1, 2, 3, 4, 8, 5, 6, 7]          # mark 8 sits between 4 and 5
acc = np.zeros(8)
ctrl = False
per_sweep = []
n = 0
for rep in range(8):
    H, sweeps, _ = eng.sparse_code(X, W, sub_iter=10, stopping_diff=0.0, seed=rep)
    buf = np.zeros(16 * 16, dtype=np.uint64)
    assert lib.onmf_debug_sweep_marks(eng.ctx, buf.ctypes.data, buf.size) > 0
    m = buf.reshape(16, 16).astype(np.int64)
    if rep < 2:
        continue
    for k in range(2, 9):
        if m[k][5] == 0:          # a kernel with a control wave: the tile waves' sweep ends at the barrier (mark 4)
            ctrl = True
            acc[:4] += np.diff(m[k][[0, 1, 2, 3, 4]]) * 0.01
            acc[4:] = 0
        else:
            acc += np.diff(m[k][ORDER]) * 0.01
        per_sweep.append((m[k + 1][0] - m[k][0]) * 0.01)
        n += 1
print("d=%d r=%d b=%d: %.2f us per sweep (workgroup 0, sweeps 2..8 of 10)" % (d, r, b, np.mean(per_sweep)))
if ctrl:
    print("  (control-wave kernel: the tile waves' sweep ends at the barrier behind the partial sums)")
for nm, v in zip(names[:4] if ctrl else names, acc / n):
    print("  %-42s %6.2f us" % (nm, v))
print("  %-42s %6.2f us" % ("to the next sweep's start", np.mean(per_sweep) - acc.sum() / n))
