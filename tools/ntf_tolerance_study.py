#!/usr/bin/env python3
"""The lasso coder's stopping tolerance (relative change per sweep and column) against its cost: sweep
counts per column, time, and how far each column's objective 0.5 |x - W h|^2 + alpha |h|_1 is from that of a
long run.  Shape of the NTF workload (d=1200, r=100), dictionary after a few NTF steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onmf_ontf_ndl_amd import _lib, Online_NTF
from onmf_ontf_ndl_amd.engine import get_engine

d, r, b = 1200, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
alpha = 2.0
eng = get_engine()
g = torch.Generator(device=eng.device).manual_seed(1234)
Xs = [torch.rand(d, b, device=eng.device, generator=g) for _ in range(3)]
W = eng.update_dict(torch.rand(d, r, device=eng.device, generator=g), eng.zeros(r, r), eng.zeros(r, d))
coder = Online_NTF(torch.zeros(2, 3, 4), n_components=r, alpha=None, device=eng.device)
A, B = eng.zeros(r, r), eng.zeros(d, r).t().contiguous()
A, B = eng.zeros(r, r), eng.zeros(r, d)
for t in range(1, 6):
    _, A, B, W = coder.step(Xs[t % 3], A, B, W, t)
X = Xs[0]
H0 = eng.zeros(r, b)


def objective(H):
    R = X.double() - W.double() @ H.double()
    return 0.5 * (R * R).sum(0) + alpha * H.double().sum(0)


Href, _ = eng.sparse_code_each(X, W, H0=H0.clone(), alpha=alpha, sub_iter=40000, stopping_diff=0.0,
                               mode=_lib.CODER_EXACT_CD, want_sweeps=True)
fref = objective(Href)
print("reference: 40000 sweeps for every column; objective mean %.6f" % fref.mean().item())
for tol in (1e-6, 2e-6, 3e-6, 5e-6, 1e-5, 3e-5, 1e-4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    H, sw = eng.sparse_code_each(X, W, H0=H0.clone(), alpha=alpha, sub_iter=4000, stopping_diff=tol,
                                 mode=_lib.CODER_EXACT_CD, want_sweeps=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    sw = sw.float()
    gap = ((objective(H) - fref) / fref)
    err = (H - Href).norm(dim=0) / Href.norm(dim=0).clamp_min(1e-30)
    print("tol %.0e: %7.2f ms; sweeps mean %6.0f median %5.0f p99 %5.0f max %5.0f, at the cap %5d; objective above the "
          "long run's: max %.2e mean %.2e; code error max %.2e median %.2e"
          % (tol, ms, sw.mean().item(), sw.median().item(), sw.quantile(0.99).item(), sw.max().item(),
             int((sw >= 4000).sum().item()), gap.max().item(), gap.mean().item(), err.max().item(), err.median().item()))
