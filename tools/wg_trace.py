#!/usr/bin/env python3
"""When and where the workgroups of a step's product kernels and of its dictionary pass ran.

Needs the diagnostic build of the library (tools/build_trace_lib.sh).  Prints, for the last of a few
steady-state steps at the given shape, every product launch with the spread of its workgroups' start
times and run times, how many of them each CU hosted, and where the dictionary pass sat meanwhile."""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from onmf_ontf_ndl_amd import _lib

_lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "libonmf_hip_trace.so")
from onmf_ontf_ndl_amd.engine import get_engine

d, r, b = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 256, 65536)
opts = [tuple(int(x) for x in a.split("=")) for a in sys.argv[4:]]
eng = get_engine()
for k, v in opts:
    eng.set_option(k, v)
lib = C.CDLL(_lib.LIB_PATH)
for name in ("onmf_debug_gemm_trace", "onmf_debug_panel_trace"):
    getattr(lib, name).restype = C.c_int
    getattr(lib, name).argtypes = [C.c_void_p, C.c_int]
g = torch.Generator(device=eng.device).manual_seed(1)
Xs = [torch.rand(d, b, device=eng.device, generator=g) for _ in range(3)]
W = eng.update_dict(torch.rand(d, r, device=eng.device, generator=g), eng.zeros(r, r), eng.zeros(r, d))
A, B = eng.zeros(r, r), eng.zeros(r, d)


def steps(n, t0):
    global A, B, W
    for t in range(n):
        _, A, B, _, W, _, _ = eng.step(Xs[t % 3], A, B, None, W, 1.0 / (t0 + t + 2), want_code=False)
    torch.cuda.synchronize()


def read(fn, cap):
    buf = np.zeros((cap, 4), dtype=np.uint64)
    n = fn(buf.ctypes.data, cap)
    assert n >= 0
    return buf[:n]


steps(12, 0)
read(lib.onmf_debug_gemm_trace, 65536)
read(lib.onmf_debug_panel_trace, 4096)
steps(3, 12)
gt = read(lib.onmf_debug_gemm_trace, 65536)
pt = read(lib.onmf_debug_panel_trace, 4096)
TICK = 0.01          # us per tick of the 100 MHz clock


def place(word):
    hw, xcc = int(word) & 0xFFFFFFFF, int(word) >> 32
    return (xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)     # XCC, SE, SH, CU


# launches: records of one tag, split where the count reaches the grid size
by_tag = collections.defaultdict(list)
for rec in gt:
    by_tag[int(rec[3])].append(rec)
launches = []
for tag, recs in by_tag.items():
    gx, gz = tag >> 32, (tag >> 20) & 0xFFF
    size = gx * gz
    recs.sort(key=lambda q: int(q[0]))
    for i in range(0, len(recs) - size + 1, size):
        launches.append((int(recs[i][0]), tag, recs[i:i + size]))
launches.sort(key=lambda q: q[0])
origin = launches[0][0]
# the last step only: from the last launch of the first tag in a step (the aggregate product, NT)
agg = [i for i, l in enumerate(launches) if (l[1] & 3) == 0 and (l[1] >> 32) * ((l[1] >> 20) & 0xFFF) > 256]
first = agg[-2] if len(agg) > 1 else 0
origin = launches[first][0]
print("shape d=%d r=%d b=%d; times in us from the start of the aggregate product before the last step" % (d, r, b))
for p in pt:
    t0, t1 = (int(p[0]) - origin) * TICK, (int(p[1]) - origin) * TICK
    if t1 < -50:
        continue
    print("dictionary pass atoms %3d..%3d: %8.1f -> %8.1f (%6.1f us) on XCC %d SE %d SH %d CU %d"
          % ((int(p[3]) >> 32, int(p[3]) & 0xFFFFFFFF, t0, t1, t1 - t0) + place(p[2])))
for t_first, tag, recs in launches[first:]:
    gx, gz = tag >> 32, (tag >> 20) & 0xFFF
    st = np.array([(int(q[0]) - origin) * TICK for q in recs])
    en = np.array([(int(q[1]) - origin) * TICK for q in recs])
    du = en - st
    print("\nproduct %s%s grid %d x %d: %8.1f -> %8.1f (%6.1f us); workgroup run time min %.1f median %.1f max %.1f"
          % ("T" if tag & 2 else "N", "T" if tag & 1 else "N", gx, gz, st.min(), en.max(), en.max() - st.min(),
             du.min(), np.median(du), du.max()))
    d1 = np.array([((int(q[2]) >> 36) & 0x3fff) * 4 * TICK for q in recs])
    d2 = np.array([((int(q[2]) >> 50) & 0x3fff) * 4 * TICK for q in recs])
    print("  inside a workgroup (median us): set-up %.2f, loads + main loop %.2f, results out %.2f"
          % (np.median(d1), np.median(d2 - d1), np.median(du - d2)))
    rel = st - st.min()
    hist = np.histogram(rel, bins=np.arange(0, rel.max() + 20, 20))[0]
    print("  starts per 20 us: " + " ".join(str(v) for v in hist))
    hist = np.histogram(en - st.min(), bins=np.arange(0, (en - st.min()).max() + 20, 20))[0]
    print("  ends   per 20 us: " + " ".join(str(v) for v in hist))
    per_cu = collections.Counter(place(q[2]) for q in recs)
    per_xcc = collections.Counter(place(q[2])[0] for q in recs)
    per_se = collections.Counter(place(q[2])[:3] for q in recs)
    print("  CUs used %d; workgroups per CU: %s" % (len(per_cu), dict(sorted(collections.Counter(per_cu.values()).items()))))
    print("  per XCC: %s" % dict(sorted(per_xcc.items())))
    print("  per (XCC,SE,SH): min %d max %d over %d" % (min(per_se.values()), max(per_se.values()), len(per_se)))
    if gx * gz >= 256:
        late = np.argsort(-st)[:6]
        for i in late:
            print("  last to start: start %7.1f end %7.1f on XCC %d SE %d SH %d CU %d" % ((st[i], en[i]) + place(recs[i][2])))
        late = np.argsort(-en)[:6]
        for i in late:
            print("  last to end: start %7.1f end %7.1f on XCC %d SE %d SH %d CU %d" % ((st[i], en[i]) + place(recs[i][2])))
