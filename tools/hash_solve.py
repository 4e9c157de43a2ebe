#!/usr/bin/env python3
"""SHA-256 of the code a solve returns at a few shapes (to compare two builds of the library bit for bit)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onmf_ontf_ndl_amd.engine import get_engine

eng = get_engine()
for d, r, b in ((1024, 256, 65536), (300, 100, 8192), (64, 16, 1000), (200, 200, 20000), (128, 144, 3000)):
    g = torch.Generator(device=eng.device).manual_seed(d + r)
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    H, sweeps, _ = eng.sparse_code(X, W, H0=H0, stopping_diff=0.01)
    print(d, r, b, sweeps, hashlib.sha256(H.cpu().numpy().tobytes()).hexdigest()[:16])
