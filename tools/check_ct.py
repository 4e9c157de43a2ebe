#!/usr/bin/env python3
"""Compare the tile sweep variants (1 / 2 / 4 column tiles per wave) with the column kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onmf_ontf_ndl_amd.engine import get_engine
from onmf_ontf_ndl_amd import _lib

eng = get_engine()
g = torch.Generator(device=eng.device).manual_seed(3)
for (d, r, b) in [(100, 16, 1000), (64, 256, 300), (300, 100, 517), (33, 7, 1), (256, 64, 2048), (128, 200, 4099),
                  (1024, 256, 65536)]:
    X = torch.rand(d, b, device=eng.device, generator=g)
    W = torch.rand(d, r, device=eng.device, generator=g)
    H0 = torch.rand(r, b, device=eng.device, generator=g)
    out = {}
    for kern in (1, 3, 4):
        if kern == 1 and b > 5000:
            continue
        eng.set_option(_lib.OPT_SWEEP_KERNEL, kern)
        for tol in (-1.0, 0.01):
            H, sweeps, _ = eng.sparse_code(X, W, H0=H0, seed=1, alpha=0.1, sub_iter=4, stopping_diff=tol)
            out[(kern, tol)] = (H.clone(), sweeps)
    ref_k = 1 if (1, -1.0) in out else 3
    for key, (H, sw) in sorted(out.items()):
        ref = out[(ref_k, key[1])][0]
        err = float((H - ref).abs().max() / ref.abs().max())
        print("shape", (d, r, b), "kernel", key[0], "tol", key[1], "sweeps", sw, "rel err vs kernel %d: %.2e" % (ref_k, err),
              flush=True)
