#!/usr/bin/env python3
"""Time the code solve (W^T W, W^T X, sweeps with the screened stopping test) at a given shape.
Usage: time_solve.py d r b [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onmf_ontf_ndl_amd.engine import get_engine

d, r, b = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
eng = get_engine()
g = torch.Generator(device=eng.device).manual_seed(1)
X = torch.rand(d, b, device=eng.device, generator=g)
W = eng.update_dict(torch.rand(d, r, device=eng.device, generator=g), eng.zeros(r, r), eng.zeros(r, d))
for _ in range(30):
    _, sweeps, _ = eng.sparse_code(X, W, seed=3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.sparse_code(X, W, seed=3)
torch.cuda.synchronize()
print("d=%d r=%d b=%d: %.1f us per solve (%d sweeps)" % (d, r, b, (time.perf_counter() - t0) / reps * 1e6, sweeps))
