#!/usr/bin/env python3
"""Sample the shader clock and the socket power (rocm-smi) while one kernel family runs back to back.
Usage: clock_under_load.py [gemm|sweep|step] [seconds]"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onmf_ontf_ndl_amd.engine import get_engine

what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
eng = get_engine()
d, r, b = 1024, 256, 65536
X = torch.rand(d, b, device=eng.device)
W = torch.rand(d, r, device=eng.device)
W = eng.update_dict(W, eng.zeros(r, r), eng.zeros(r, d))
H = torch.rand(r, b, device=eng.device)
A0 = torch.zeros(r, r, device=eng.device)
B0 = torch.zeros(r, d, device=eng.device)
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True,
                                 timeout=10).stdout
            keep = [l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "mclk" in l]
            samples.append(" | ".join(keep))
        except Exception as e:        # noqa: BLE001 - a sampling tool: report and go on
            samples.append("rocm-smi failed: %r" % (e,))
        time.sleep(0.3)


def work():
    if what == "gemm":
        eng.aggregate(X, H, 0.5, A0, B0)
    elif what == "sweep":
        eng.sparse_code(X, W, H0=H, sub_iter=10, stopping_diff=0.0)
    else:
        eng.step(X, A0, B0, None, W, 0.5, H0=H)


work()
torch.cuda.synchronize()
th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(20):
        work()
    torch.cuda.synchronize()
    n += 20
el = time.time() - t0
stop = True
th.join()
print("%s: %d calls in %.2f s = %.1f us per call" % (what, n, el, el / n * 1e6))
for s in samples:
    print(s)
