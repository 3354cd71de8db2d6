#!/usr/bin/env python3
"""Per-wavefront kernel durations of ONE device FMT* solve out of a rocprofv3 rocpd database (the run of tools/run_wavefront_steps.py):
usage: wavefront_steps.py results.db [which solve, counted from 0]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(cur.execute("select %s, start, end from kernels order by start" % name_col))
inits = [i for i, r in enumerate(rows) if "k_wf_init(" in r[0]]
a = inits[which]; b = inits[which + 1] if which + 1 < len(inits) else len(rows)
seq = [(r[0].split("(")[0].replace("void ", "").split("<")[0], (r[2] - r[1]) / 1e3, r[1]) for r in rows[a:b] if "k_wf_" in r[0]]
step, out = [], []
for n, d, st in seq:
    if n == "k_wf_apply_min" and step: out.append(step); step = []
    step.append((n, d, st))
out.append(step)
print("wavefront  apply select   mark compact connect   gaps  (us)")
tot = {}
for i, s in enumerate(out):
    dd = {n: d for n, d, _ in s}
    if "k_wf_connect" not in dd: continue
    span = (s[-1][2] - s[0][2]) / 1e3 + s[-1][1]
    ksum = sum(d for _, d, _ in s)
    for k, v in dd.items(): tot[k] = tot.get(k, 0.0) + v
    print("%6d    %6.1f %6.1f %6.1f %6.1f %7.1f %6.1f" % (i, dd.get("k_wf_apply_min", 0), dd.get("k_wf_select", 0), dd.get("k_wf_mark", 0), dd.get("k_wf_compact", 0), dd.get("k_wf_connect", 0), span - ksum))
print("sums (us):", {k: round(v, 1) for k, v in tot.items()})
