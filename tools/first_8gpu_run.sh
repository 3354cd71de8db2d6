#!/bin/bash
# The FIRST run on a real multi-GPU node (no round of this build has had one): bench.py at 1, 2, 4 and 8 ranks over real RCCL, the
# per-rank diagnostics of every run, and a consistency check of the sharded results against the single-GPU run --
#   * the shards' nnz add up to the unsharded graph's (same sample set: --sample-sets 1),
#   * a checksum of the gathered free-edge mask: the number of free edges over all shards equals the unsharded mask's popcount
#     (bench.py prints both in `consistency`),
#   * the scaling table (ms per step, edges/s, efficiency against N = 1) from the driver's own JSON lines.
# usage: bash tools/first_8gpu_run.sh [outdir]        (one node, 8 visible GPUs; ~2 minutes)
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/first_8gpu}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 1 2 4 8; do
  echo "== $n rank(s) =="
  timeout 900 python bench.py --gpus $n --steps 20 --warmup 5 --sample-sets 1 --no-cpu-baseline --no-solve --no-cold > "$OUT/bench_n$n.json" 2> "$OUT/bench_n$n.err" \
    || { echo "bench.py --gpus $n failed:"; tail -20 "$OUT/bench_n$n.err"; }
done
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
rows = {}
for n in (1, 2, 4, 8):
    p = os.path.join(out, "bench_n%d.json" % n)
    try:
        rows[n] = json.loads([l for l in open(p) if l.startswith("{")][-1])
    except Exception as e:
        print("N=%d: no JSON line (%s)" % (n, e))
if 1 in rows:
    base = rows[1]
    print("%-3s %10s %14s %10s %12s %s" % ("N", "ms/step", "edges/s", "speed-up", "efficiency", "nnz == N=1 ?"))
    for n, d in sorted(rows.items()):
        sp = base["ms_per_step"] / d["ms_per_step"]
        print("%-3d %10.3f %14.4g %10.2f %11.0f%% %s" % (n, d["ms_per_step"], d["value"], sp, 100 * sp / n,
                                                       "ok" if d["config"]["nnz"] == base["config"]["nnz"] else "MISMATCH %d vs %d" % (d["config"]["nnz"], base["config"]["nnz"])))
    for n, d in sorted(rows.items()):
        pr = d.get("per_rank")
        if pr:
            print("N=%d per rank: step call %.3f..%.3f ms, pair kernel %.3f..%.3f, ordering %.3f..%.3f, exposed gather wait %.3f..%.3f ms" % (
                n, pr["step_call_ms"]["min"], pr["step_call_ms"]["max"], pr["pair_kernel"]["min"], pr["pair_kernel"]["max"],
                pr["rdisc_sort"]["min"], pr["rdisc_sort"]["max"], pr["gather_exposed_ms"]["min"], pr["gather_exposed_ms"]["max"]))
        c = d.get("consistency")
        if c and 1 in rows and rows[1].get("consistency"):
            ok = c["free_edges"] == rows[1]["consistency"]["free_edges"]
            print("N=%d free edges over all shards %d -- %s" % (n, c["free_edges"], "equal to N=1" if ok else "MISMATCH (N=1: %d)" % rows[1]["consistency"]["free_edges"]))
PY
