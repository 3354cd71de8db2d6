#!/bin/bash
# Round-end evidence for the build in this tree (run on the GPU box through gpurun; everything under its own timeout):
#   profiles/${R}_bench_ns_${V}.json              python bench.py (the driver's command)
#   profiles/${R}_bench_ns_kernel_stats_${V}.csv  rocprofv3 --kernel-trace --stats of the same command (per-kernel durations)
#   profiles/${R}_pmc_${V}.txt + profiles/traffic.json   rocprofv3 --pmc passes (separate run: no trace domains beside the counters)
# usage: bash tools/final_profiles.sh r02 v2     (outputs are also copied to gpurun_out/ so they travel back)
R=${1:-rXX}; V=${2:-v0}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python3 bench.py > profiles/${R}_bench_ns_${V}.json 2> /tmp/bench.err || { tail -5 /tmp/bench.err; exit 1; }
NNZ=$(python3 -c "import json;print(json.load(open('profiles/${R}_bench_ns_${V}.json'))['config']['nnz'])")
rm -rf /tmp/prof_stats /tmp/pmc
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o s -- python3 $OLDPWD/bench.py --no-cpu-baseline --no-cold --no-solve --steps 10 > /tmp/stats.log 2>&1)
DB=$(find /tmp/prof_stats -name "*_results.db" | head -1)
[ -n "$DB" ] && python3 tools/rocpd_stats.py "$DB" profiles/${R}_bench_ns_kernel_stats_${V}.csv > /dev/null
(cd /tmp && timeout 300 rocprofv3 -i $OLDPWD/tools/pmc_traffic.txt --kernel-trace --output-format csv -d /tmp/pmc -o p -- python3 $OLDPWD/bench.py --no-cpu-baseline --no-solve --no-cold --steps 3 --warmup 1 > /tmp/pmc.log 2>&1)
python3 tools/pmc_traffic.py /tmp/pmc profiles/${R}_pmc_${V}.txt ns_r6_n1m_m200 $NNZ > /dev/null
timeout 600 python3 bench.py > profiles/${R}_bench_ns_${V}.json 2> /tmp/bench.err      # again, now that traffic.json matches this build
cp profiles/${R}_bench_ns_${V}.json profiles/${R}_bench_ns_kernel_stats_${V}.csv profiles/${R}_pmc_${V}.txt profiles/traffic.json gpurun_out/ 2>/dev/null
python3 - <<PY
import json
d = json.load(open("profiles/${R}_bench_ns_${V}.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "traffic", "valu_frac")},
      "solve", d["submetrics"].get("fmt_solve", {}).get("ms"))
PY
