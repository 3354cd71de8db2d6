#!/bin/bash
# A/B of two builds on the same box, alternating: build_ab/libmpfmt_A.so vs build_ab/libmpfmt_B.so
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for v in A B; do
  printf "%s " $v
  MPFMT_LIB_PATH=$PWD/build_ab/libmpfmt_$v.so timeout 200 python bench.py --no-cpu-baseline --no-solve --steps 30 ${WL:+--workload $WL} 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['submetrics']['kernel_ms']; print(round(k['rdisc_count'],3), round(k['sweep_graph'],3), round(k['rdisc_sort'],3), round(d['ms_per_step'],3))"
done; done
