"""Screened vs exact stopping test on one solve: sweeps run and the per-sweep dist record.

    python tools/trace_dist.py [d r b] [steps]
With steps > 0 the solve is repeated on a carried dictionary (as the bench does) and the per-step
sweep-kernel launches are printed."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from onmf_ontf_ndl_amd.engine import get_engine
from onmf_ontf_ndl_amd import _lib
eng = get_engine()
g = torch.Generator(device=eng.device).manual_seed(1)
d, r, b = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 256, 65536)
steps = int(sys.argv[4]) if len(sys.argv) >= 5 else 0
X = torch.rand(d, b, device=eng.device, generator=g)
W = torch.rand(d, r, device=eng.device, generator=g)
W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
H, sweeps, tr = eng.sparse_code(X, W, H0=None, seed=1, alpha=0.0, sub_iter=10, stopping_diff=0.01)
print("screened", sweeps, tr)
eng.set_option(_lib.OPT_EXACT_STOP, 1)
H, sweeps, tr = eng.sparse_code(X, W, H0=None, seed=1, alpha=0.0, sub_iter=10, stopping_diff=0.01)
print("exact", sweeps, tr)
eng.set_option(_lib.OPT_EXACT_STOP, 0)
if steps:
    A, B = eng.zeros(r, r), eng.zeros(r, d)
    W = eng.update_dict(W, A, B)
    eng.set_option(_lib.OPT_PROFILE, 1)
    for t in range(1, steps + 1):
        eng.get_profile()
        _, A, B, _, W, s, tr = eng.step(X, A, B, None, W, 1.0 / t, alpha=0.0, H0=None, seed=t, sub_iter=10,
                                        stopping_diff=0.01, want_code=False)
        ms, n = eng.get_profile()
        print(t, "sweeps", s, "launches", n, "ms %.3f" % ms, ["%.4f" % v for v in tr])
