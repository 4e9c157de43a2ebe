#!/usr/bin/env python3
"""Time the dictionary pass alone (both kernels) at a given shape."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--r", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    from onmf_ontf_ndl_amd.engine import get_engine
    from onmf_ontf_ndl_amd import _lib
    eng = get_engine()
    g = torch.Generator(device=eng.device).manual_seed(1)
    W = torch.rand(args.d, args.r, device=eng.device, generator=g)
    Hh = torch.rand(args.r, 4096, device=eng.device, generator=g)
    A = Hh @ Hh.T / 4096
    B = Hh @ torch.rand(args.d, 4096, device=eng.device, generator=g).T / 4096
    for kern in (1, 2):
        eng.set_option(_lib.OPT_DICT_KERNEL, kern)
        eng.update_dict(W, A, B)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            W1 = eng.update_dict(W, A, B)
        e1.record()
        torch.cuda.synchronize()
        print("dict kernel %d: %.3f ms per pass (d=%d r=%d), checksum %.6f" %
              (kern, e0.elapsed_time(e1) / args.reps, args.d, args.r, float(W1.sum())), flush=True)


if __name__ == "__main__":
    main()
