"""Kernel timeline of the last step in a rocprofv3 kernel-trace DB (gaps between kernels = host round trips / launch latency).
usage: step_timeline.py results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'k_graph_sweep' in r[0]]
a, b = idx[-2] + 1, idx[-1]
t0 = rows[a][1]; prev_end = rows[a - 1][2]; tot = 0
for r in rows[a:b + 1]:
    print("%8.1f us  +%6.1f gap  dur %7.1f  %s" % ((r[1] - t0) / 1e3, (r[1] - prev_end) / 1e3, (r[2] - r[1]) / 1e3, r[0][:60]))
    prev_end = r[2]; tot += r[2] - r[1]
print("step span %.1f us, kernel sum %.1f us, launches %d" % ((rows[b][2] - rows[idx[-2]][2]) / 1e3, tot / 1e3, b - a + 1))
