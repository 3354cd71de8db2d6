#!/bin/bash
# Kernel timeline of one north-star bench step (on the GPU box): start, duration, gap of every kernel -> stdout
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/prof_tl
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_tl -o s -- python3 $OLDPWD/bench.py --no-cpu-baseline --no-cold --no-solve --steps 6 --warmup 2 > /tmp/tl.log 2>&1)
DB=$(find /tmp/prof_tl -name "*_results.db" | head -1)
python3 tools/step_timeline.py $DB 3
