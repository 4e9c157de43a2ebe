#!/usr/bin/env python3
"""Run the library's two big products a few times (for a rocprofv3 --pmc pass)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onmf_ontf_ndl_amd.engine import get_engine

eng = get_engine()
d, r, b = 1024, 256, 65536
X = torch.rand(d, b, device=eng.device)
W = torch.rand(d, r, device=eng.device)
H = torch.rand(r, b, device=eng.device)
A0 = torch.zeros(r, r, device=eng.device)
B0 = torch.zeros(r, d, device=eng.device)
for _ in range(3):
    eng.aggregate(X, H, 0.5, A0, B0)                      # H X^T and H H^T  (gemm <false,false>)
    eng.sparse_code(X, W, H0=H, sub_iter=0)               # W^T W and W^T X  (gemm <true,true>)
torch.cuda.synchronize()
