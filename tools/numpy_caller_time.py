#!/usr/bin/env python3
"""What a NumPy caller of the class surface pays per call: Online_NMF.step with float64 NumPy arrays in and
out (the reference's calling convention), against the same step on resident device tensors."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from onmf_ontf_ndl_amd import Online_NMF
from onmf_ontf_ndl_amd.engine import get_engine

eng = get_engine()
for d, r, b, n in ((100, 16, 1000, 60), (1024, 64, 8192, 30), (1024, 256, 65536, 6)):
    rs = np.random.RandomState(0)
    Xs = [rs.rand(d, b) for _ in range(3)]          # fresh batches in turn, as a stream brings them
    X = Xs[0]
    W = rs.rand(d, r)
    A, B = np.zeros((r, r)), np.zeros((r, d))
    nmf = Online_NMF(X, n_components=r, iterations=2, batch_size=b)
    H, agg, W1 = nmf.step(X, [A, B], W, 1)          # warm-up (allocations, first launches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        H, agg, W1 = nmf.step(Xs[t % 3], agg, W1, t + 2)
    host = (time.perf_counter() - t0) / n
    nmf_dev = Online_NMF(X, n_components=r, iterations=2, batch_size=b, rng="device", seed=1)
    nmf_dev.step(X, agg, W1, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        H, agg, W1 = nmf_dev.step(Xs[t % 3], agg, W1, t + 2)
    host_devrng = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    h0 = np.random.rand(r, b)
    draw = time.perf_counter() - t0
    nmf = nmf_dev
    Xds = [eng.dev(x) for x in Xs]
    Xd, Wd = Xds[0], eng.dev(W1)
    Ad, Bd = eng.dev(agg[0]), eng.dev(agg[1])
    nmf.step(Xd, [Ad, Bd], Wd, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        Hd, aggd, Wd = nmf.step(Xds[t % 3], [Ad, Bd], Wd, t + 2)
        Ad, Bd = aggd
    torch.cuda.synchronize()
    devt = (time.perf_counter() - t0) / n
    # device tensors in and out, H0 from NumPy's own stream (drawn on the device out of NumPy's state)
    nmf_np = Online_NMF(X, n_components=r, iterations=2, batch_size=b, rng="numpy")
    nmf_np.step(Xd, [Ad, Bd], Wd, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        Hd, aggd, Wd = nmf_np.step(Xds[t % 3], [Ad, Bd], Wd, t + 2)
        Ad, Bd = aggd
    torch.cuda.synchronize()
    devt_np = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    h0d = eng.numpy_rand(r, b)
    torch.cuda.synchronize()
    draw_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    x32 = torch.from_numpy(X).to(eng.device, torch.float32)
    torch.cuda.synchronize()
    up = time.perf_counter() - t0
    t0 = time.perf_counter()
    h64 = Hd.cpu().numpy().astype(np.float64)
    down = time.perf_counter() - t0
    print("d=%d r=%d b=%d: NumPy float64 in/out, H0 from NumPy's generator (the reference's stream) %.2f ms per step "
          "(%.2f M patches/s); the same with rng='device' %.2f ms; device tensors in/out, rng='device' %.2f ms; "
          "device tensors in/out, rng='numpy' (NumPy's stream continued on the device) %.2f ms; "
          "parts: np.random.rand(r, b) on the host %.2f ms, the same draw on the device %.2f ms, X up %.2f ms, "
          "H down and to float64 %.2f ms"
          % (d, r, b, host * 1e3, b / host / 1e6, host_devrng * 1e3, devt * 1e3, devt_np * 1e3, draw * 1e3,
             draw_dev * 1e3, up * 1e3, down * 1e3))
