#!/usr/bin/env python3
"""Compare the screened stopping test's per-sweep bounds with the exact dist values."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--r", type=int, default=256)
    ap.add_argument("--b", type=int, default=65536)
    ap.add_argument("--trained", type=int, default=0, help="dictionary updates before the comparison")
    args = ap.parse_args()
    import torch
    from onmf_ontf_ndl_amd import _lib
    from onmf_ontf_ndl_amd.engine import get_engine
    eng = get_engine()
    g = torch.Generator(device=eng.device).manual_seed(1)
    X = torch.rand(args.d, args.b, device=eng.device, generator=g)
    W = torch.rand(args.d, args.r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    A, B = eng.zeros(args.r, args.r), eng.zeros(args.r, args.d)
    for t in range(args.trained):
        _, A, B, _, W, _, _ = eng.step(X, A, B, None, W, 1.0 / (t + 1), seed=100 + t, stopping_diff=-1.0)
    H0 = eng.rand(args.r, args.b, seed=5)
    eng.set_option(_lib.OPT_EXACT_STOP, 1)
    _, s_exact, d_exact = eng.sparse_code(X, W, H0=H0, stopping_diff=0.01)
    eng.set_option(_lib.OPT_EXACT_STOP, 0)
    _, s_scr, d_scr = eng.sparse_code(X, W, H0=H0, stopping_diff=0.01)
    print("sweeps exact %d screened %d" % (s_exact, s_scr))
    for i in range(max(len(d_exact), len(d_scr))):
        e = d_exact[i] if i < len(d_exact) else float("nan")
        s = d_scr[i] if i < len(d_scr) else float("nan")
        print("sweep %2d exact dist %.5f  screened value %.5f  ratio %.3f" % (i, e, s, s / e if e else 0))


if __name__ == "__main__":
    main()
