#!/usr/bin/env python3
"""Run a few sparse-code solves (fixed 10 sweeps) at a given shape - a short target for rocprofv3."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--r", type=int, default=256)
    ap.add_argument("--b", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--tol", type=float, default=0.0)
    ap.add_argument("--kernel", type=int, default=0, help="ONMF_OPT_SWEEP_KERNEL value")
    args = ap.parse_args()
    import torch
    from onmf_ontf_ndl_amd.engine import get_engine
    eng = get_engine()
    from onmf_ontf_ndl_amd import _lib
    eng.set_option(_lib.OPT_SWEEP_KERNEL, args.kernel)
    g = torch.Generator(device=eng.device).manual_seed(1)
    X = torch.rand(args.d, args.b, device=eng.device, generator=g)
    W = torch.rand(args.d, args.r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    for i in range(args.reps):
        H, sweeps, _ = eng.sparse_code(X, W, H0=None, seed=i + 1, alpha=0.0, sub_iter=10,
                                       stopping_diff=args.tol)
    torch.cuda.synchronize()
    print("sweeps", sweeps, "H mean", float(H.mean()))


if __name__ == "__main__":
    main()
