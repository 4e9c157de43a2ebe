import sys
sys.path.insert(0, '/root/repo')
import torch
from onmf_ontf_ndl_amd.engine import get_engine
from onmf_ontf_ndl_amd import _lib
eng = get_engine()
g = torch.Generator(device=eng.device).manual_seed(1)
d, r, b = 1024, 256, 65536
X = torch.rand(d, b, device=eng.device, generator=g)
W = torch.rand(d, r, device=eng.device, generator=g)
W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
H, sweeps, tr = eng.sparse_code(X, W, H0=None, seed=1, alpha=0.0, sub_iter=10, stopping_diff=0.01)
print("screened", sweeps, tr)
eng.set_option(_lib.OPT_EXACT_STOP, 1)
H, sweeps, tr = eng.sparse_code(X, W, H0=None, seed=1, alpha=0.0, sub_iter=10, stopping_diff=0.01)
print("exact", sweeps, tr)
