#!/usr/bin/env python3
"""Which stopping tests of a stream of steps needed a full evaluation, with and without the device-side
evaluation (ONMF_OPT_DEVICE_RESOLVE): per step the sweep count, the reported values (a lower bound where
the bounds decided, the exact value where the test was evaluated in full) and the counters.
    python3 tools/studies/small_shape_tests.py d r b [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from onmf_ontf_ndl_amd.engine import get_engine

d, r, b = (int(a) for a in sys.argv[1:4])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
eng = get_engine()
for opt9 in (0, 1):
    eng.set_option(9, opt9)
    g = torch.Generator(device=eng.device).manual_seed(3)
    Xs = [torch.rand(d, b, device=eng.device, generator=g) for _ in range(3)]
    W = torch.rand(d, r, device=eng.device, generator=g)
    A, B = eng.zeros(r, r), eng.zeros(r, d)
    print("device-side evaluation", "on" if opt9 else "off")
    for t in range(1, steps + 1):
        before = eng.counters()
        _, A, B, _, W, sweeps, trace = eng.step(Xs[t % 3], A, B, None, W, 1.0 / t, seed=100 + t)
        after = eng.counters()
        print("  step %2d: %2d sweeps, device evaluations %d, values %s" % (
            t, sweeps, after["device_resolves"] - before["device_resolves"], " ".join("%.5f" % v for v in trace)))
eng.set_option(9, 0)
