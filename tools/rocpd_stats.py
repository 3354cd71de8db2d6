#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (SQLite) result file as a per-kernel stats table (the equivalent of
`--stats` CSV output): calls, total / average / min / max duration.  Usage: rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    q = ("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
         "from kernels group by %s order by sum(end-start) desc" % (name_col, name_col))
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for r in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
