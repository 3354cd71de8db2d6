#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, per counter: sum over dispatches / number of dispatches."""
import csv, glob, sys, collections
root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if filt and filt not in k:
            continue
        k = k[:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        n = cnt[k][c]
        print("   %-34s %18.1f  (per dispatch, %d dispatches)" % (c, acc[k][c] / n, n))
