#!/bin/bash
# Instruction counts of the sweep kernel for the stage-ablated builds (build_abl/, tools/ablate_sweep.sh): one rocprofv3 --pmc pass each.
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
for n in ${ABLS:-0 2 3 4}; do
  rm -rf /tmp/pmci$n
  (cd /tmp && MPFMT_OPT_SWEEP_SORTED=${SORTED:-1} MPFMT_LIB_PATH=$R/build_abl/libmpfmt_abl$n.so timeout 300 rocprofv3 -i $R/tools/pmc_insts.txt --kernel-trace --output-format csv -d /tmp/pmci$n -o p -- python3 $R/bench.py --no-cpu-baseline --no-solve --steps 3 --warmup 1 > /tmp/pmci$n.log 2>&1)
  python3 - $n <<'PY'
import csv, glob, sys, collections
n = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('/tmp/pmci%s/**/*counter_collection.csv' % n, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_graph_sweep' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print('SWEEP_ABL=%s ' % n + '  '.join('%s=%.4g' % (k, v[0] / max(v[1], 1)) for k, v in sorted(acc.items())))
PY
done
