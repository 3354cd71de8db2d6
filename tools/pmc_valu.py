#!/usr/bin/env python3
"""Turn a rocprofv3 --pmc run of `bench.py --workload cfg4 | cfg5` into an entry of profiles/valu_ops.json, the file those bench legs
read the vector-instruction counts of their `roofline` from (the kernels are bound by vector-ALU issue: achieved = SQ_INSTS_VALU x 64
lanes per second against 39.3e12 lane-ops/s).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 -i tools/pmc_valu.txt --kernel-trace --output-format csv -d /tmp/pmc4 -o p -- python3 bench.py --workload cfg4 --no-cpu-baseline --steps 2 --warmup 1
  python3 tools/pmc_valu.py /tmp/pmc4 profiles/rNN_pmc_cfg4.txt cfg4_di_r4_n100000 k_di_pairs k_di_sweep

Per kernel name (substring) the LARGEST dispatch class is taken (the pilot pass of the double-integrator count runs the same kernel on
every 32nd tile: by SQ_INSTS_VALU it is 1/32 of the real launch), averaged over its dispatches.  Keyed to the sha256 of libmpfmt.so:
bench.py prints null for a stale entry."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", "valu_ops.json")


def main():
    src, out_txt, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    subs = sys.argv[4:]
    rows = collections.defaultdict(dict)          # (kernel, dispatch id) -> counter -> value
    for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            rows[(row["Kernel_Name"], row.get("Dispatch_Id") or row.get("Correlation_Id") or str(len(rows)))][row["Counter_Name"]] = float(row["Counter_Value"])
    per = collections.defaultdict(list)
    for (k, _), c in rows.items():
        per[k].append(c)
    lines = []
    sha = hashlib.sha256(open(os.path.join(ROOT, "motionplanning.jl_amd", "libmpfmt.so"), "rb").read()).hexdigest()
    entry = {"lib_sha256": sha, "source": os.path.relpath(out_txt, ROOT)}
    for k in sorted(per, key=lambda k: -sum(c.get("SQ_INSTS_VALU", 0) for c in per[k])):
        ds = per[k]
        big = max(c.get("SQ_INSTS_VALU", 0) for c in ds)
        top = [c for c in ds if c.get("SQ_INSTS_VALU", 0) >= 0.5 * big] or ds
        avg = {n: sum(c.get(n, 0) for c in top) / len(top) for n in sorted(set().union(*[set(c) for c in top]))}
        lines.append("%s   (%d dispatches, %d in the largest class)" % (k[:110], len(ds), len(top)))
        for n, v in avg.items():
            lines.append("   %-34s %20.1f" % (n, v))
        for sub in subs:
            if sub in k and avg.get("SQ_INSTS_VALU", 0) > entry.get(sub, 0):
                entry[sub] = avg["SQ_INSTS_VALU"]
                if avg.get("SQ_ACTIVE_INST_VALU") and avg.get("GRBM_GUI_ACTIVE"):
                    entry[sub + "_valu_busy"] = avg["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (avg["GRBM_GUI_ACTIVE"] / 8.0)
    os.makedirs(os.path.dirname(os.path.abspath(out_txt)), exist_ok=True)
    open(out_txt, "w").write("\n".join(lines) + "\n")
    try:
        allw = json.load(open(OUT))
    except Exception:
        allw = {}
    allw[workload] = entry
    json.dump(allw, open(OUT, "w"), indent=1, sort_keys=True)
    print(json.dumps(entry))


if __name__ == "__main__":
    main()
