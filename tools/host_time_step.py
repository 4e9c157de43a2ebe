#!/usr/bin/env python3
"""Where does the host's time go in a short step?  Wall time per Engine.step call, the part spent inside the
library call, and the GPU time of the same steps (events)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onmf_ontf_ndl_amd.engine import get_engine

d, r, b = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 64, 8192)
eng = get_engine()
g = torch.Generator(device=eng.device).manual_seed(1)
Xs = [torch.rand(d, b, device=eng.device, generator=g) for _ in range(3)]
W = eng.update_dict(torch.rand(d, r, device=eng.device, generator=g), eng.zeros(r, r), eng.zeros(r, d))
A, B = eng.zeros(r, r), eng.zeros(r, d)
inner = [0.0]
real = eng.lib.onmf_step


def timed_step(*a):
    t0 = time.perf_counter()
    rc = real(*a)
    inner[0] += time.perf_counter() - t0
    return rc


class Lib:
    def __getattr__(self, name):
        return timed_step if name == "onmf_step" else getattr(eng_lib, name)


eng_lib = eng.lib
eng.lib = Lib()
for t in range(30):
    _, A, B, _, W, _, _ = eng.step(Xs[t % 3], A, B, None, W, 1.0 / (t + 2))
torch.cuda.synchronize()
inner[0] = 0.0
n = 200
t0 = time.perf_counter()
for t in range(n):
    _, A, B, _, W, _, _ = eng.step(Xs[t % 3], A, B, None, W, 1.0 / (t + 32))
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("d=%d r=%d b=%d: %.1f us per step, %.1f us of it inside onmf_step, %.1f us in Python around it"
      % (d, r, b, wall / n * 1e6, inner[0] / n * 1e6, (wall - inner[0]) / n * 1e6))

# The same with steps that are not waited for (ONMF_OPT_DEFER_VERDICT): the call returns once everything is
# enqueued; the next call first reads the previous call's verdicts.  And what the enqueueing alone costs: a
# call on an idle, settled context waits for nothing.
from onmf_ontf_ndl_amd import _lib
eng.set_option(_lib.OPT_DEFER_VERDICT, 1)
for t in range(30):
    _, A, B, _, W, _, _ = eng.step(Xs[t % 3], A, B, None, W, 1.0 / (t + 2), want_code=False, defer=True)
eng.settle()
torch.cuda.synchronize()
inner[0] = 0.0
t0 = time.perf_counter()
for t in range(n):
    _, A, B, _, W, _, _ = eng.step(Xs[t % 3], A, B, None, W, 1.0 / (t + 32), want_code=False, defer=True)
eng.settle()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("not waited for: %.1f us per step, %.1f us of it inside onmf_step (enqueueing and the wait for the previous "
      "step's verdicts)" % (wall / n * 1e6, inner[0] / n * 1e6))
alone = []
for t in range(20):
    eng.settle()
    torch.cuda.synchronize()
    inner[0] = 0.0
    _, A, B, _, W, _, _ = eng.step(Xs[t % 3], A, B, None, W, 1.0 / (t + 300), want_code=False, defer=True)
    alone.append(inner[0])
eng.settle()
torch.cuda.synchronize()
alone.sort()
print("enqueueing one step on an idle context: median %.1f us (min %.1f)" % (alone[len(alone) // 2] * 1e6, alone[0] * 1e6))
