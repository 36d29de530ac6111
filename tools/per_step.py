#!/usr/bin/env python3
"""Kernel durations per wavefront step of the last render in a rocprofv3 kernel trace of a one-stream run
(tools/trace1.sh): tools/per_step.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
start = [i for i, k in enumerate(ks) if "k_init_slots" in k[2]][-1]
seg = ks[start:]
t0 = seg[0][0]
out, cur = [], {}
for s, e, n in seg:
    short = ("adv" if "k_advance" in n else "cam" if "k_camera" in n else "odd" if "intersect_odd" in n else
             "shadow" if "k_shadow" in n else "walk" if "k_intersect" in n else None)
    if short is None:
        continue
    if short == "adv" and cur:
        out.append(cur)
        cur = {}
    cur[short] = (e - s) / 1e3
    cur.setdefault("t", (s - t0) / 1e6)
    cur["end"] = (e - t0) / 1e6
out.append(cur)
for i, c in enumerate(out):
    busy = sum(v for k, v in c.items() if k not in ("t", "end"))
    print(i, {k: round(v, 1) for k, v in c.items()}, "idle %.0f us" % ((c["end"] - c["t"]) * 1e3 - busy))
