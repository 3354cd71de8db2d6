#!/bin/bash
# round 5, first GPU call: parity suite, bench line, per-rank shard simulation
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5_pytest1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_pytest1.log
tail -15 gpurun_out/r5_pytest1.log
timeout 300 python bench.py > gpurun_out/r5_bench1.json 2> gpurun_out/r5_bench1.err; tail -3 gpurun_out/r5_bench1.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r5_bench1.json"))
    print("ms_per_step", d["ms_per_step"], {k: v for k, v in d.get("submetrics", {}).items() if k in ("timers", "kernel_ms")})
except Exception as e:
    print("bench parse failed", e)
PY
timeout 600 python tools/run_shard_sim.py > gpurun_out/r5_shard_sim1.txt 2>&1; cat gpurun_out/r5_shard_sim1.txt
