import sys
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from onmf_ontf_ndl_amd.engine import get_engine
eng = get_engine()
d, r, b = 64, 16, 500
g = torch.Generator(device=eng.device).manual_seed(d + r)
X = torch.rand(d, b, device=eng.device, generator=g)
W = torch.rand(d, r, device=eng.device, generator=g)
W = W / torch.clamp(torch.linalg.norm(W, dim=0), min=1.0)
H0 = torch.rand(r, b, device=eng.device, generator=g)
eng.set_option(2, 1)
He, se, de = eng.sparse_code(X, W, H0=H0, stopping_diff=0.01)
eng.set_option(2, 0)
Hs, ss, ds = eng.sparse_code(X, W, H0=H0, stopping_diff=0.01)
print(se, de); print(ss, ds)
# exact sigma values per sweep by replaying one sweep at a time
H = H0.clone()
for i in range(se):
    Hn, _, _ = eng.sparse_code(X, W, H0=H, sub_iter=1, stopping_diff=-1.0) if i == 0 else (None, None, None)
    break
