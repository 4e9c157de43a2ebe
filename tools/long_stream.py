#!/usr/bin/env python3
"""Drift of the learned dictionary over a long stream: the GPU path (fp32) against the NumPy oracle
(fp64) on the same patch stream and the same random starts, step by step.

    python tools/long_stream.py [--d 100 --r 16 --b 1000 --steps 100]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=100)
    ap.add_argument("--r", type=int, default=16)
    ap.add_argument("--b", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    from onmf_ontf_ndl_amd.engine import get_engine
    from oracle import onmf_oracle as orc
    eng = get_engine()
    d, r, b = args.d, args.r, args.b
    rs = np.random.RandomState(5)
    W = rs.rand(d, r)
    A, B = np.zeros((r, r)), np.zeros((r, d))
    Wd, Ad, Bd = eng.dev(W), eng.dev(A), eng.dev(B)
    worst, flips = 0.0, 0
    for t in range(1, args.steps + 1):
        X = rs.rand(d, b)
        H0 = rs.rand(r, b)
        tr = []
        _, (A, B), W, _ = orc.minibatch_step(X, [A, B], W, t, H0=H0, trace=tr, norm="gram")
        _, Ad, Bd, _, Wd, sweeps, _ = eng.step(eng.dev(X), Ad, Bd, None, Wd, 1.0 / t, H0=eng.dev(H0))
        flips += int(sweeps != len(tr))
        err = np.linalg.norm(Wd.cpu().numpy() - W) / np.linalg.norm(W)
        worst = max(worst, err)
        if t % 10 == 0 or t == 1:
            print("step %3d  sweeps gpu %2d oracle %2d  W rel err %.2e  A rel err %.2e" % (
                t, sweeps, len(tr), err, np.linalg.norm(Ad.cpu().numpy() - A) / np.linalg.norm(A)), flush=True)
    print("worst W rel err %.2e, steps with a different sweep count: %d" % (worst, flips))


if __name__ == "__main__":
    main()
