#!/usr/bin/env python3
"""Per-kernel register/LDS/scratch table from `hipcc -Rpass-analysis=kernel-resource-usage` output on stdin."""
import re
import sys

cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("sgpr", r" SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    if flt in r["name"]:
        print("%-70s sgpr %3s vgpr %3s scratch %4s occ %s" % (r["name"][-70:], r.get("sgpr"), r.get("vgpr"),
                                                              r.get("scratch"), r.get("occ")))
