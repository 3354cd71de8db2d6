#!/usr/bin/env python3
"""Timeline of a cold device solve out of a rocprofv3 rocpd database (tools/cold_solve.py with one context): every kernel and copy
up to the end of the first solve whose gap to the previous one exceeds 30 us, and per-phase sums.  Usage: cold_timeline.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = [(n, s, e) for n, s, e in cur.execute("select name, start, end from kernels order by start")]
try:
    rows += [("memcpy " + str(n), s, e) for n, s, e in cur.execute("select name, start, end from memory_copies order by start")]
except Exception as ex:
    print("(no memory copies in the trace:", ex, ")")
rows.sort(key=lambda r: r[1])
paths = [i for i, r in enumerate(rows) if "k_wf_path" in r[0]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # which context of the run (three solves each)
begin = 0 if which == 0 else paths[3 * which - 1] + 1
while which and "memcpy" not in rows[begin][0]: begin += 1
first_path = paths[3 * which]
t0 = rows[begin][1]; prev = t0; busy = 0.0
print("%-60s %10s %9s %9s" % ("event", "start us", "dur us", "gap us"))
for n, s, e in rows[begin:first_path + 1]:
    gap = (s - prev) / 1e3
    if gap > 30 or (e - s) > 100e3:
        print("%-60s %10.1f %9.1f %9.1f" % (n[:60], (s - t0) / 1e3, (e - s) / 1e3, gap))
    busy += (e - s) / 1e3; prev = max(prev, e)
print("span %.1f us, device busy %.1f us" % ((prev - t0) / 1e3, busy))
