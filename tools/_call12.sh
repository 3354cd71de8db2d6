cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROOT=$(pwd)
cat > /tmp/wf.py <<PY
import sys, os, time
sys.path.insert(0, "$ROOT")
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
ctx = mp.Context(0)
ctx.upload_samples(w.X); ctx.upload_boxes(w.lohi, w.ss_lo, w.ss_hi)
ctx.graph_step_device(w.r)
for _ in range(2):
    res = ctx.fmtstar_wavefront(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r, want_tree=False)
# per-step batch / candidate counts through the step-wise API
ctx.wf_begin(w.r, mp._lib.GOAL_BALL, w.goal_params(), band=0.25 * w.r)
rows = []
while True:
    info = ctx.wf_step()
    rows.append((info["nz"], info["nx"], info["nconn"]))
    if info["done"]: break
print("STEPS", rows)
PY
rm -rf /tmp/prof_wf
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_wf -o s -- python3 /tmp/wf.py > /tmp/wfp.log 2>&1)
grep STEPS /tmp/wfp.log | head -2
DB=$(find /tmp/prof_wf -name "*_results.db" | head -1)
python3 - <<PY
import sqlite3
db = sqlite3.connect("$DB"); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(cur.execute("select %s, start, end from kernels order by start" % name_col))
# the second automatic solve: find k_wf_init occurrences
inits = [i for i, r in enumerate(rows) if "k_wf_init(" in r[0]]
a, b = inits[1], inits[2]
seq = [(r[0].split("(")[0].replace("void ", "").split("<")[0], (r[2] - r[1]) / 1e3, r[1]) for r in rows[a:b] if "k_wf_" in r[0]]
step = []; out = []
for n, d, st in seq:
    if n == "k_wf_apply_min" and step: out.append(step); step = []
    step.append((n, d, st))
out.append(step)
print("step  apply select  mark compact connect   gap_total  (us)")
for i, s in enumerate(out):
    dd = {n: d for n, d, _ in s}
    if "k_wf_connect" not in dd: continue
    span = (s[-1][2] - s[0][2]) / 1e3 + s[-1][1]
    ksum = sum(d for _, d, _ in s)
    print("%3d  %6.1f %6.1f %6.1f %6.1f %7.1f   %6.1f" % (i, dd.get("k_wf_apply_min", 0), dd.get("k_wf_select", 0), dd.get("k_wf_mark", 0), dd.get("k_wf_compact", 0), dd.get("k_wf_connect", 0), span - ksum))
PY
