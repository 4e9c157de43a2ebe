#!/usr/bin/env python3
"""How tight would a Rayleigh-Ritz lower bound on 16 orthonormal probes be, against the per-probe bound?
Offline study on the library's own iterates (codes after k = 0..10 sweeps from one start)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--r", type=int, default=64)
    ap.add_argument("--b", type=int, default=8192)
    ap.add_argument("--trained", type=int, default=6)
    args = ap.parse_args()
    import torch
    from onmf_ontf_ndl_amd.engine import get_engine
    eng = get_engine()
    g = torch.Generator(device=eng.device).manual_seed(1)
    W = torch.rand(args.d, args.r, device=eng.device, generator=g)
    W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
    A, B = eng.zeros(args.r, args.r), eng.zeros(args.r, args.d)
    for t in range(args.trained):
        X = torch.rand(args.d, args.b, device=eng.device, generator=g)
        _, A, B, _, W, _, _ = eng.step(X, A, B, None, W, 1.0 / (t + 1), seed=100 + t, stopping_diff=-1.0)
    U = None
    for trial in range(6):
        X = torch.rand(args.d, args.b, device=eng.device, generator=g)
        H0 = eng.rand(args.r, args.b, seed=5 + trial)
        Hs = [H0.double()]
        for k in range(1, 11):
            H, _, _ = eng.sparse_code(X, W, H0=H0, sub_iter=k, stopping_diff=-1.0)
            Hs.append(H.double().clone())
        if U is None:
            U = torch.linalg.qr(torch.randn(args.r, 16, device=eng.device, dtype=torch.float64, generator=None))[0]
        print("trial %d" % trial)
        margins = []
        for k in range(10):
            D = Hs[k + 1] - Hs[k]
            if k == 1 and trial % 4 == 0:  # the refresh (every fourth batch): one power step on the move of the second sweep
                U = torch.linalg.qr(D @ (D.T @ U))[0]
            s1 = torch.linalg.matrix_norm(D, 2).item()
            P = U.T @ D
            per_probe = P.pow(2).sum(1).max().sqrt().item()
            ritz = torch.linalg.eigvalsh(P @ P.T)[-1].sqrt().item()
            s0 = torch.linalg.matrix_norm(Hs[k], 2).item()
            fD = torch.linalg.matrix_norm(D).item()
            if k == 2:
                ub = s0                      # a tight bound of sigma_max of the code before the third sweep
            if k >= 2:
                infl = ub / s0
                ub = ub + fD                 # the bound the next sweep inherits
            else:
                infl = float("nan")
            if k >= 1:
                margins.append(per_probe * s1 / s0 / 0.01)
            print("  sweep %d dist %.5f  per-probe/exact %.3f  ritz/exact %.4f  ||dH||_F/sigma %.2f  inherited bound of sigma(H)/exact %.3f"
                  % (k, s1 / s0, per_probe / s1, ritz / s1, fD / s1, infl))
        print("  lower bound / tol with an exact sigma(H), sweeps 1..9:", " ".join("%.3f" % m for m in margins))


if __name__ == "__main__":
    main()
