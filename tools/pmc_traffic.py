#!/usr/bin/env python3
"""Turn a rocprofv3 --pmc run of bench.py into profiles/traffic.json, the file bench.py reads its `roofline.traffic` from.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 -i tools/pmc_traffic.txt --kernel-trace --output-format csv -d /tmp/pmc -o p -- python3 bench.py --no-cpu-baseline --no-solve --steps 3 --warmup 1
  python3 tools/pmc_traffic.py /tmp/pmc profiles/rNN_pmc_<tag>.txt          (writes the per-kernel summary there and profiles/traffic.json)

traffic (bytes per launch) follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half
the bytes of a wide coalesced read, so it is doubled -- the guide's correction.  That correction is NOT established for gathers
(tools/ubench/fetch_calib.hip: FETCH_SIZE = 64 B x L2 read requests for random 48-byte rows), so `bytes_gather_calibrated`
(FETCH_SIZE x 1 + WRITE_SIZE) is stored beside it; the truth lies between the two for kernels that mix streams and gathers.
The summary is keyed to the exact library build (sha256 of libmpfmt.so), the workload and the shard count: bench.py reports
traffic = null when any of them differs."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_sha():
    return hashlib.sha256(open(os.path.join(ROOT, "motionplanning.jl_amd", "libmpfmt.so"), "rb").read()).hexdigest()


def main():
    src, out_txt = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "ns_r6_n1m_m200"
    nnz = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
    per = {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}
    lines = []
    for k in sorted(per, key=lambda k: -per[k].get("SQ_BUSY_CYCLES", 0)):
        lines.append(k[:100])
        for c in sorted(per[k]):
            lines.append("   %-34s %18.1f  (per dispatch, %d dispatches)" % (c, per[k][c], cnt[k][c]))
    open(out_txt, "w").write("\n".join(lines) + "\n")

    def pick(sub):
        ks = [k for k in per if sub in k]
        return per[max(ks, key=lambda k: per[k].get("SQ_BUSY_CYCLES", 0))] if ks else None
    out = {"lib_sha256": lib_sha(), "workload": workload, "n_gpus": 1, "source": os.path.relpath(out_txt, ROOT),
           "method": "bytes = (FETCH_SIZE x 2 + WRITE_SIZE) KiB x 1024 (MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE is half the bytes of a "
                     "coalesced read); bytes_gather_calibrated = (FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gathers: FETCH_SIZE = 64 B x read requests)"}
    for tag, sub in (("pair", "k_rdisc_mfma"), ("sweep", "k_graph_sweep"), ("sort", "k_order_logs"), ("exact", "k_exact_pairs"), ("pending", "k_sweep_pending")):
        p = pick(sub)
        if not p:
            continue
        e = {"fetch_size_kib": p.get("FETCH_SIZE"), "write_size_kib": p.get("WRITE_SIZE"),
             "bytes": (p.get("FETCH_SIZE", 0) * 2 + p.get("WRITE_SIZE", 0)) * 1024,
             "bytes_gather_calibrated": (p.get("FETCH_SIZE", 0) + p.get("WRITE_SIZE", 0)) * 1024,
             "insts_valu": p.get("SQ_INSTS_VALU"), "insts_salu": p.get("SQ_INSTS_SALU"), "insts_mfma": p.get("SQ_INSTS_MFMA"),
             "l2_hit_rate": (p.get("TCC_HIT_sum", 0) / max(p.get("TCC_HIT_sum", 0) + p.get("TCC_MISS_sum", 0), 1))}
        if p.get("SQ_INSTS_MFMA"):
            e["valu_per_mfma"] = p["SQ_INSTS_VALU"] / p["SQ_INSTS_MFMA"]
        if p.get("SQ_WAVE_CYCLES"):
            e["wait_frac"] = p.get("SQ_WAIT_ANY", 0) / p["SQ_WAVE_CYCLES"]          # wave cycles parked in s_waitcnt / barriers
        if p.get("SQ_ACTIVE_INST_VALU") and p.get("GRBM_GUI_ACTIVE"):
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE cycles summed over the 8 XCDs
            e["valu_busy"] = p["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (p["GRBM_GUI_ACTIVE"] / 8.0)
        if p.get("SQ_VALU_MFMA_BUSY_CYCLES") and p.get("GRBM_GUI_ACTIVE"):
            # matrix-pipe busy cycles summed over the 1024 SIMDs over the kernel's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
            e["mfma_busy"] = p["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (p["GRBM_GUI_ACTIVE"] / 8.0)
        if tag == "sweep" and nnz:
            e["valu_lane_ops_per_edge"] = p.get("SQ_INSTS_VALU", 0) * 64.0 / nnz
        out[tag] = e
    # the whole step: every kernel's traffic, per step (steps = dispatches of the pair kernel in the run)
    pk = [k for k in per if "k_rdisc_mfma" in k]
    if pk:
        steps = max(cnt[max(pk, key=lambda k: per[k].get("SQ_BUSY_CYCLES", 0))].get("FETCH_SIZE", 0), 1)
        by = {}
        for k in per:
            b = (acc[k].get("FETCH_SIZE", 0.0) * 2 + acc[k].get("WRITE_SIZE", 0.0)) * 1024 / steps
            if b > 0:
                by[k.split("(")[0][:60]] = by.get(k.split("(")[0][:60], 0.0) + b
        out["step"] = {"steps_in_run": steps, "bytes": sum(by.values()),
                       "by_kernel": {k: v for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]}}
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
