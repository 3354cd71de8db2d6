#!/bin/bash
# Quick counter pass of the bench step (run on the GPU box): instruction mix, waits, LDS conflicts, L2 traffic of every kernel.
# usage: bash tools/pmc_quick.sh <tag> [counter file]      -> gpurun_out/pmc_<tag>.txt
TAG=${1:-x}; CF=${2:-tools/pmc_quick.txt}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/pmcq
(cd /tmp && timeout 400 rocprofv3 -i $OLDPWD/$CF --kernel-trace --output-format csv -d /tmp/pmcq -o p -- python3 $OLDPWD/bench.py --no-cpu-baseline --no-solve --no-cold --steps 3 --warmup 1 > /tmp/pmcq.log 2>&1)
python3 tools/pmc_summary.py /tmp/pmcq > gpurun_out/pmc_${TAG}.txt
