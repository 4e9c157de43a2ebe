#!/usr/bin/env python3
"""How fast is the vendor fp32 GEMM (through torch.matmul) on the step's three products, next to this library's?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)
d, r, b = 1024, 256, 65536
X = torch.rand(d, b, device=dev)
W = torch.rand(d, r, device=dev)
H = torch.rand(r, b, device=dev)


def timeit(fn, flops, name, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-28s %.3f ms  %.1f TFLOP/s" % (name, ms, flops / ms / 1e9), flush=True)


timeit(lambda: W.t() @ X, 2.0 * d * r * b, "vendor  W^T X")
timeit(lambda: H @ X.t(), 2.0 * d * r * b, "vendor  H X^T")
timeit(lambda: H @ H.t(), 2.0 * r * r * b, "vendor  H H^T")
from onmf_ontf_ndl_amd.engine import get_engine
eng = get_engine()
A0 = torch.zeros(r, r, device=dev)
B0 = torch.zeros(r, d, device=dev)
timeit(lambda: eng.aggregate(X, H, 0.5, A0, B0), 2.0 * d * r * b + 2.0 * r * r * b, "library H X^T + H H^T")
timeit(lambda: eng.sparse_code(X, W, H0=H, sub_iter=0), 2.0 * d * r * b + 2.0 * d * r * r, "library W^T X + W^T W")
