#!/usr/bin/env python3
"""Sweeps, verdict traces and wall time of the first few solves of a fresh context (probe cold start)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onmf_ontf_ndl_amd.engine import get_engine

eng = get_engine()
g = torch.Generator(device=eng.device).manual_seed(1)
d, r, b = 1024, 256, 65536
X = torch.rand(d, b, device=eng.device, generator=g)
W = torch.rand(d, r, device=eng.device, generator=g)
W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
for i in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    H, sweeps, tr = eng.sparse_code(X, W, H0=None, seed=i + 1, alpha=0.0, sub_iter=10, stopping_diff=0.01)
    torch.cuda.synchronize()
    print("solve %d: %d sweeps, %.2f ms, trace %s" % (i, sweeps, 1e3 * (time.perf_counter() - t0),
                                                      " ".join("%.4f" % v for v in tr)), flush=True)
