#!/bin/bash
# One library option over several values on the same box (bench.py reads MPFMT_OPT_<NAME>):
#   bash tools/opt_sweep.sh mf_xcd_mode "512 64 8"        (WL=cfg2 for another workload)
cd "$(dirname "$0")/.."
OPT=$1; UP=$(echo $OPT | tr a-z A-Z)
for m in $2; do
  printf "%s %-7s " $OPT $m
  env MPFMT_OPT_$UP=$m timeout 300 python bench.py --no-cpu-baseline --no-solve --no-cold --steps 30 ${WL:+--workload $WL} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['submetrics']['kernel_ms']; print('grid %.3f pair %.3f exact %.3f sort %.3f step %.3f' % (k['grid'], k['pair_kernel']-k['exact_pairs'], k['exact_pairs'], k['rdisc_sort'], d['ms_per_step']))"
done
