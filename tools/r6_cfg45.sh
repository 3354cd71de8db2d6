#!/bin/bash
# bench lines + counters of BASELINE configs[3] (cfg4, double integrator) and configs[4] (cfg5, Monte Carlo): profiles/${R}_bench_cfg4.json, _cfg5.json,
# per-kernel stats and the SQ_INSTS_VALU summaries profiles/valu_ops.json is made from.   usage: bash tools/r6_cfg45.sh r06
R=${1:-rXX}
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
for wl in cfg4 cfg5; do
  rm -rf /tmp/pmc_$wl /tmp/st_$wl
  (cd /tmp && timeout 600 rocprofv3 -i $ROOT/tools/pmc_valu.txt --kernel-trace --output-format csv -d /tmp/pmc_$wl -o p -- python3 $ROOT/bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 > /tmp/pmc_$wl.log 2>&1) || tail -3 /tmp/pmc_$wl.log
  if [ $wl = cfg4 ]; then python3 tools/pmc_valu.py /tmp/pmc_$wl profiles/${R}_pmc_$wl.txt cfg4_di_r4_n100000 k_di_pairs k_di_sweep
  else python3 tools/pmc_valu.py /tmp/pmc_$wl profiles/${R}_pmc_$wl.txt cfg5_mc_r6_m200 k_mc_edges k_mc_ais_edges; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/st_$wl -o s -- python3 $ROOT/bench.py --workload $wl --no-cpu-baseline --steps 3 --warmup 1 > /tmp/st_$wl.log 2>&1)
  DB=$(find /tmp/st_$wl -name "*_results.db" | head -1)
  [ -n "$DB" ] && python3 tools/rocpd_stats.py "$DB" profiles/${R}_bench_${wl}_kernel_stats.csv > /dev/null
  timeout 900 python3 bench.py --workload $wl > profiles/${R}_bench_$wl.json 2> /tmp/b_$wl.err || tail -5 /tmp/b_$wl.err
  head -c 900 profiles/${R}_bench_$wl.json; echo
done
cp profiles/${R}_bench_cfg4* profiles/${R}_bench_cfg5* profiles/${R}_pmc_cfg4.txt profiles/${R}_pmc_cfg5.txt profiles/valu_ops.json gpurun_out/ 2>/dev/null
