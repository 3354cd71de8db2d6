#!/bin/bash
# Per-kernel durations of the bench step (run on the GPU box): rocprofv3 --kernel-trace --stats -> gpurun_out/kstats_<tag>.csv
TAG=${1:-x}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/prof_stats
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o s -- python3 $OLDPWD/bench.py --no-cpu-baseline --no-solve --no-cold --steps 10 > /tmp/stats.log 2>&1)
DB=$(find /tmp/prof_stats -name "*_results.db" | head -1)
[ -n "$DB" ] && python3 tools/rocpd_stats.py "$DB" gpurun_out/kstats_${TAG}.csv > /dev/null
