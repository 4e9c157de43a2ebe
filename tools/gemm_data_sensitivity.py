#!/usr/bin/env python3
"""Time the two big products on operands of different content (uniform random, mostly zero, all zero).
The arithmetic is the same; a difference in duration is the clock responding to the power drawn."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onmf_ontf_ndl_amd.engine import get_engine

eng = get_engine()
d, r, b = 1024, 256, 65536
A0 = torch.zeros(r, r, device=eng.device)
B0 = torch.zeros(r, d, device=eng.device)


def timed(fn, reps=50):
    import time
    t0 = time.time()
    while time.time() - t0 < 0.6:            # the clock ramps up from its idle level in the first 0.3 s of load
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, dens in (("random", 1.0), ("30% nonzero", 0.3), ("zeros", 0.0)):
    X = torch.rand(d, b, device=eng.device)
    W = torch.rand(d, r, device=eng.device)
    H = torch.rand(r, b, device=eng.device)
    if dens < 1.0:
        H = H * (torch.rand(r, b, device=eng.device) < dens)
        if dens == 0.0:
            X.zero_()
            W.zero_()
    t_agg = timed(lambda: eng.aggregate(X, H, 0.5, A0, B0))
    t_code = timed(lambda: eng.sparse_code(X, W, H0=H, sub_iter=0))
    print("%-12s aggregates %.1f us   W^T W + W^T X %.1f us" % (name, t_agg, t_code), flush=True)
