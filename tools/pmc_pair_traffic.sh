#!/bin/bash
# Counter traffic of the three main kernels for several values of one library option (on the GPU box; ~2.5 min per value):
#   bash tools/pmc_pair_traffic.sh mf_xcd_mode "256 512"
OPT=$1; UP=$(echo $OPT | tr a-z A-Z)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
cp profiles/traffic.json /tmp/traffic_keep.json
for m in $2; do
  rm -rf /tmp/pmc; export MPFMT_OPT_$UP=$m
  (cd /tmp && timeout 300 rocprofv3 -i $OLDPWD/tools/pmc_traffic.txt --kernel-trace --output-format csv -d /tmp/pmc -o p -- python3 $OLDPWD/bench.py --no-cpu-baseline --no-solve --no-cold --steps 3 --warmup 1 > /tmp/pmc.log 2>&1)
  python3 tools/pmc_traffic.py /tmp/pmc /tmp/pmc_$m.txt ns_r6_n1m_m200 107492200 > /dev/null
  python3 -c "
import json; d=json.load(open('profiles/traffic.json'))
print('$OPT $m', ' | '.join('%s fetch %.0f KiB write %.0f KiB bytes %.3e' % (k, d[k]['fetch_size_kib'], d[k]['write_size_kib'], d[k]['bytes']) for k in ('pair','sort','exact')), 'step %.3e' % d['step']['bytes'])"
done
cp /tmp/traffic_keep.json profiles/traffic.json
