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

# ---- round 5: the other lines DESIGN.md quotes (each under its own timeout; failures do not stop the rest) ----
ROOT=$(pwd)
for wl in cfg2 cfg3; do
  timeout 900 python3 bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 3 > profiles/${R}_bench_${wl}.json 2> /tmp/b_$wl.err || tail -3 /tmp/b_$wl.err
done
timeout 600 python3 bench.py --workload cfg1 --n 400000 --no-cpu-baseline --no-cold --no-solve --steps 20 > profiles/${R}_bench_cfg1_n400k.json 2> /tmp/b_c1.err || tail -3 /tmp/b_c1.err
# round 6: BASELINE configs[3] / configs[4] with their counters (profiles/valu_ops.json is keyed to this build), and the clustered north star
bash tools/r6_cfg45.sh ${R} > /tmp/cfg45.log 2>&1 || tail -5 /tmp/cfg45.log
timeout 600 python3 bench.py --workload ns_clustered --no-cpu-baseline > profiles/${R}_bench_clustered.json 2> /tmp/b_cl.err || tail -3 /tmp/b_cl.err
timeout 900 python3 tools/run_shard_sim.py > profiles/${R}_shard_sim.txt 2>&1
timeout 900 python3 tools/run_shard_ab.py 8 > profiles/${R}_shard_ab_g8.txt 2>&1
timeout 1500 python3 tools/run_form_grid.py > profiles/${R}_form_grid.txt 2>&1
for spec in "3 8" "0 1"; do
  set -- $spec
  rm -rf /tmp/prof_sh
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_sh -o s -- python3 $ROOT/tools/run_shard_one.py $1 $2 > /tmp/sh.log 2>&1)
  DB=$(find /tmp/prof_sh -name "*_results.db" | head -1)
  [ -n "$DB" ] && python3 tools/step_timeline.py $DB 2 > profiles/${R}_step_timeline_g$2_rank$1.txt 2>&1
done
# the device FMT* solve on the resident north-star graph: per-kernel durations and counters
rm -rf /tmp/prof_wf /tmp/pmc_wf
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_wf -o s -- python3 $ROOT/tools/run_wavefront_ns.py > $ROOT/profiles/${R}_wavefront_ns.txt 2>&1)
DB=$(find /tmp/prof_wf -name "*_results.db" | head -1)
[ -n "$DB" ] && python3 tools/rocpd_stats.py "$DB" profiles/${R}_wavefront_kernel_stats_${V}.csv > /dev/null
(cd /tmp && timeout 600 rocprofv3 -i $ROOT/tools/pmc_mem.txt --kernel-trace --output-format csv -d /tmp/pmc_wf -o p -- python3 $ROOT/tools/run_wavefront_ns.py > /tmp/pmc_wf.log 2>&1)
python3 tools/pmc_summary.py /tmp/pmc_wf k_wf > profiles/${R}_pmc_wavefront_${V}.txt 2>&1
cp profiles/${R}_* gpurun_out/ 2>/dev/null
rm -rf /tmp/prof_ws
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_ws -o s -- python3 $ROOT/tools/run_wavefront_steps.py > /tmp/ws.log 2>&1)
DB=$(find /tmp/prof_ws -name "*_results.db" | head -1)
[ -n "$DB" ] && { grep "per wavefront" /tmp/ws.log; python3 tools/wavefront_steps.py $DB 1; } > profiles/${R}_wavefront_steps.txt 2>&1
cp profiles/${R}_* gpurun_out/ 2>/dev/null
