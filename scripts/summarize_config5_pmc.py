#!/usr/bin/env python3
"""Summarise scripts/profile_config5_pmc.sh's counter runs: python scripts/summarize_config5_pmc.py <gpurun_out/tag> <out.json> [storage]
(an existing <out.json> keeps its legs of other storages: one file holds the F32-arithmetic and the split-arithmetic pass)

Per FULL-batch launch of the pass (64 pairs; the dispatch that applies fewer pairs -- none in this run -- would be dropped by its
grid): HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB; MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE tallies the 128-byte requests of
16-byte-per-lane streaming reads at 64 bytes, WRITE_SIZE is exact for 16-byte-per-lane stores), matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES
summed over 1 024 SIMDs / GRBM_GUI_ACTIVE summed over 8 XCDs, L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)."""
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
storage = sys.argv[3] if len(sys.argv) > 3 else "f32_mixed"
KERNELS = ("k_flush_strip32", "k_flush_mfma32", "k_flush_split3<")


def per_dispatch(name, counter):
    files = glob.glob(os.path.join(src, "pmc_" + name, "*", "*_counter_collection.csv"))
    if not files:
        return {}, None
    acc, kern = {}, None
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in KERNELS):
            acc[r["Dispatch_Id"]] = acc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            kern = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ekf_pipe32::", "")
    return acc, kern


def avg(d):
    return sum(d.values()) / len(d) if d else None


fetch, kern = per_dispatch("fetch", "FETCH_SIZE")
write, _ = per_dispatch("write", "WRITE_SIZE")
busy, _ = per_dispatch("mfma", "SQ_VALU_MFMA_BUSY_CYCLES")
act, _ = per_dispatch("mfma", "GRBM_GUI_ACTIVE")
hit, _ = per_dispatch("l2", "TCC_HIT_sum")
miss, _ = per_dispatch("l2", "TCC_MISS_sum")
line = None
for name in ("fetch", "stats"):
    try:
        line = json.loads([ln for ln in open(os.path.join(src, name + ".json")) if ln.startswith("{")][-1])
        break
    except (OSError, IndexError, ValueError):
        pass
leg = {"landmarks": 40000, "batch": 64, "storage": storage, "kernel": kern, "pairs_per_launch": 64,
       "dispatches": {"fetch": len(fetch), "write": len(write), "mfma": len(busy), "l2": len(hit)}}
if line:
    leg["workload"] = line["config"]["workload"]
    leg["algorithmic_bytes_per_launch"] = line["roofline"]["algorithmic_bytes_per_launch"]
    leg["avg_launch_ms_under_counters"] = line["roofline"]["avg_launch_ms"]
if fetch and write:
    leg["FETCH_SIZE_KiB_avg"], leg["WRITE_SIZE_KiB_avg"] = avg(fetch), avg(write)
    leg["hbm_bytes_per_launch"] = 2 * avg(fetch) * 1024 + avg(write) * 1024
    if line:
        leg["traffic_over_algorithmic"] = leg["hbm_bytes_per_launch"] / leg["algorithmic_bytes_per_launch"]
if busy and act:
    leg["matrix_pipe_busy"] = (avg(busy) / 1024.0) / (avg(act) / 8.0)
    leg["SQ_VALU_MFMA_BUSY_CYCLES_avg"], leg["GRBM_GUI_ACTIVE_avg"] = avg(busy), avg(act)
if hit and miss:
    leg["TCC_HIT_sum_avg"], leg["TCC_MISS_sum_avg"] = avg(hit), avg(miss)
    leg["l2_hit_rate"] = avg(hit) / (avg(hit) + avg(miss))
ks = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
if ks:
    for r in csv.DictReader(open(ks[0])):
        if any(k in r["Name"] for k in KERNELS):
            leg["rocprofv3_kernel_stats"] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6,
                                            "max_ms": float(r["MaxNs"]) / 1e6}
out = {"correction": "gfx950: FETCH_SIZE x2 (128-B requests tallied at 64 B, 16-B/lane streaming reads); WRITE_SIZE exact (MI355X_MICROARCH.md, HBM)",
       "source": "scripts/profile_config5_pmc.sh: rocprofv3 --pmc <one group per run> --kernel-include-regex '<the pass>' on "
                 "scripts/bench_config5.py --landmarks 40000 --steps 192 --warmup 64 --batch 64 --storage <leg's storage>",
       "legs": [leg]}
try:
    old = json.load(open(dst))
    out["legs"] = [l for l in old.get("legs", []) if l.get("storage") != storage] + [leg]
except (OSError, ValueError):
    pass
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
