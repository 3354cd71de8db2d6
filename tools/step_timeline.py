#!/usr/bin/env python3
"""One step's kernel timeline out of a rocprofv3 rocpd database: start offset, duration and the gap to the previous kernel's end.
Usage: step_timeline.py results.db [which_step_from_the_end]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(cur.execute("select %s, start, end from kernels order by start" % name_col))
# a step starts at its k_cellkey (the index build); the read-back of the step before it comes first in the listing
starts = [i for i, r in enumerate(rows) if "k_cellkey" in r[0]]
a, b = starts[-back - 1], starts[-back]
t0 = rows[a][1]; prev_end = t0
print("%-58s %10s %10s %9s" % ("kernel", "start us", "dur us", "gap us"))
for n, s, e in rows[a:b]:
    print("%-58s %10.1f %10.1f %9.1f" % (n[:58], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3)); prev_end = e
print("step span %.1f us; next step's first kernel starts %.1f us after this step's last ends" % ((prev_end - t0) / 1e3, (rows[b][1] - prev_end) / 1e3 if b < len(rows) else -1))
