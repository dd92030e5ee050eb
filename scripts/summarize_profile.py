#!/usr/bin/env python3
"""Turn a gpurun_out/<tag>/ profile directory (scripts/profile_round.sh) into the tracked summaries under profiles/:
    python scripts/summarize_profile.py <tag> <roundN>
writes profiles/<roundN>_bench.json, _bench_under_rocprof.json, _kernel_stats.csv, _downdate_pmc.json (the file bench.py reads).

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB and come from separate
--pmc passes; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced
streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import csv
import glob
import json
import os
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]          # e.g. r2p round2
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % rnd))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "%s_bench.json" % rnd))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "%s_bench_under_rocprof.json" % rnd))


def counter_by_kernel(kind, counter, kernel_substr):
    """Average counter value per dispatch, per kernel instance name (template arguments included)."""
    f = glob.glob(os.path.join(src, "pmc_%s" % kind, "*", "*_counter_collection.csv"))[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in kernel_substr) and r["Counter_Name"] == counter:
            acc.setdefault((r["Kernel_Name"], int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


bench = json.loads(open(os.path.join(src, "bench.json")).readline())
N, tile = bench["config"]["landmarks"], bench["config"]["tile"]
b_alg = bench["roofline"]["algorithmic_bytes_per_launch"]
dbatch = bench.get("deferred", {}).get("deferred_batch")
fetch = counter_by_kernel("fetch", "FETCH_SIZE", ("k_downdate", "k_flush"))
write = counter_by_kernel("write", "WRITE_SIZE", ("k_downdate", "k_flush"))
legs = []
for (kname, grid), (f_kib, nf) in sorted(fetch.items(), key=lambda kv: kv[0][1]):
    w_kib, nw = write[(kname, grid)]
    short = next((k for k in ("k_flush_mfma", "k_flush_lds", "k_downdate_w", "k_downdate") if k in kname), kname[:40])
    pairs = 1 if short.startswith("k_downdate") else dbatch           # the immediate leg launches the one-pair kernel
    for key in ("deferred", ) + tuple(k for k in bench if k.startswith("deferred_b")):
        # several deferred legs (batch 20, batch 32): the leg whose launcher reported this kernel instance
        inst = bench.get(key, {}).get("roofline", {}).get("kernel", "")
        if inst and inst.replace(" ", "").rstrip(">") in kname.replace(" ", ""):
            pairs = bench[key]["deferred_batch"]
    rec = {"kernel": short, "kernel_instance": kname.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0],
           "grid_size": grid, "landmarks": N, "tile": tile, "batch": pairs,
           "FETCH_SIZE_KiB_avg": f_kib, "fetch_dispatches": nf, "WRITE_SIZE_KiB_avg": w_kib, "write_dispatches": nw,
           "hbm_read_bytes_per_launch": 2.0 * f_kib * 1024.0, "hbm_write_bytes_per_launch": w_kib * 1024.0,
           "hbm_bytes_per_launch": 2.0 * f_kib * 1024.0 + w_kib * 1024.0, "algorithmic_bytes_per_launch": b_alg}
    rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / b_alg
    legs.append(rec)
mf = glob.glob(os.path.join(src, "pmc_mfma", "*", "*_counter_collection.csv"))
if mf:
    busy, act = {}, {}                               # per kernel instance: dispatch -> counter value
    for r in csv.DictReader(open(mf[0])):
        if "k_flush_mfma" in r["Kernel_Name"]:
            d = busy if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES" else act if r["Counter_Name"] == "GRBM_GUI_ACTIVE" else None
            if d is not None:
                dd = d.setdefault(r["Kernel_Name"], {})
                dd[r["Dispatch_Id"]] = dd.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    for leg in legs:
        for kname in busy:
            if leg["kernel"] == "k_flush_mfma" and leg["kernel_instance"] in kname.replace("(anonymous namespace)::", "") and kname in act:
                bq = sum(busy[kname].values()) / len(busy[kname])
                aq = sum(act[kname].values()) / len(act[kname])
                # busy cycles summed over 1024 SIMDs / active cycles summed over 8 XCDs
                leg["matrix_pipe_busy"] = (bq / 1024.0) / (aq / 8.0)
out = {"correction": "gfx950: FETCH_SIZE x2 (128-B requests tallied at 64 B, 16-B/lane streaming reads); WRITE_SIZE exact "
                     "(MI355X_MICROARCH.md, HBM)",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py, %s" % tag,
       "legs": legs}
json.dump(out, open(os.path.join(dst, "%s_downdate_pmc.json" % rnd), "w"), indent=1)
print(json.dumps(out, indent=1))
