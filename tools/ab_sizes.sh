#!/bin/bash
# Two or more builds (build_ab/libmpfmt_<NAME>.so) over several sample counts of the north-star world:
#   SIZES="500000 640000 2000000" bash tools/ab_sizes.sh A B
cd "$(dirname "$0")/.."
for n in ${SIZES:-500000 640000 1000000 2000000}; do for v in "$@"; do
  printf "N %-8s %-8s " $n $v
  MPFMT_LIB_PATH=$PWD/build_ab/libmpfmt_$v.so timeout 300 python bench.py --no-cpu-baseline --no-solve --no-cold --steps 20 --n $n 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['submetrics']['kernel_ms']; print('tiles %d slices %d pair %.3f exact %.3f sort %.3f step %.3f' % (d['submetrics']['tiles'], d['submetrics']['slices'], k['pair_kernel']-k['exact_pairs'], k['exact_pairs'], k['rdisc_sort'], d['ms_per_step']))"
done; done
