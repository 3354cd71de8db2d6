#!/bin/bash
# A/B/C... of several builds on the same box, alternating: every build_ab/libmpfmt_<NAME>.so named on the command line
#   bash tools/ab_multi.sh A B C        (WL=cfg2 for another workload, REPS=3)
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${REPS:-3}); do for v in "$@"; do
  printf "%-10s " $v
  MPFMT_LIB_PATH=$PWD/build_ab/libmpfmt_$v.so timeout 300 python bench.py --no-cpu-baseline --no-solve --no-cold --steps 30 ${WL:+--workload $WL} 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['submetrics']['kernel_ms']; print('grid %.3f pair %.3f exact %.3f sort %.3f count %.3f sweep %.3f step %.3f ord_per_cu %s' % (k['grid'], k['pair_kernel']-k['exact_pairs'], k['exact_pairs'], k['rdisc_sort'], k['rdisc_count'], k['sweep_kernel'], d['ms_per_step'], d['submetrics'].get('launch', {}).get('ord_per_cu')))"
done; done
