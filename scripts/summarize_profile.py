#!/usr/bin/env python3
"""Turn a gpurun_out/<tag>/ profile directory (scripts/profile_round.sh) into the tracked summaries under profiles/.

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

tag, rnd = sys.argv[1], sys.argv[2]          # e.g. r01a round1
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % rnd))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "%s_bench.json" % rnd))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "%s_bench_under_rocprof.json" % rnd))


def counter_by_grid(kind, counter, kernel_substr):
    """Average counter value per dispatch, split by the kernel's grid size (the batched flush and the one-pair
    downdate are the same kernel template launched with different slabs / grids)."""
    f = glob.glob(os.path.join(src, "pmc_%s" % kind, "*", "*_counter_collection.csv"))[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in kernel_substr) and r["Counter_Name"] == counter:
            acc.setdefault((r["Kernel_Name"], int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


bench = json.load(open(os.path.join(src, "bench.json")))
fetch = counter_by_grid("fetch", "FETCH_SIZE", ("k_downdate", "k_flush"))
write = counter_by_grid("write", "WRITE_SIZE", ("k_downdate", "k_flush"))
legs = []
for (kname, grid), (f_kib, nf) in sorted(fetch.items(), key=lambda kv: kv[0][1]):
    w_kib, nw = write[(kname, grid)]
    # which leg: the immediate kernel covers 4 rows of a tile per workgroup, the batched flush 32 or 64 rows -> 8x / 16x fewer workgroups
    immediate = grid == max(g for (_, g) in fetch)
    batch = 1 if immediate else bench["config"]["deferred_batch"]
    rec = {"kernel": next((k for k in ("k_flush_mfma", "k_flush_lds", "k_downdate_w") if k in kname), kname[:40]), "grid_size": grid, "landmarks": bench["config"]["landmarks"],
           "tile": bench["config"]["tile"], "batch": batch,
           "FETCH_SIZE_KiB_avg": f_kib, "fetch_dispatches": nf, "WRITE_SIZE_KiB_avg": w_kib, "write_dispatches": nw,
           "hbm_read_bytes_per_launch": 2.0 * f_kib * 1024.0, "hbm_write_bytes_per_launch": w_kib * 1024.0,
           "hbm_bytes_per_launch": 2.0 * f_kib * 1024.0 + w_kib * 1024.0,
           "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]}
    rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    legs.append(rec)
out = {"correction": "gfx950: FETCH_SIZE x2 (128-B requests tallied at 64 B, 16-B/lane streaming reads); WRITE_SIZE exact "
                     "(MI355X_MICROARCH.md, HBM)",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py, %s" % tag,
       "legs": legs}
json.dump(out, open(os.path.join(dst, "downdate_pmc.json"), "w"), indent=1)
shutil.copy(os.path.join(dst, "downdate_pmc.json"), os.path.join(dst, "%s_downdate_pmc.json" % rnd))
print(json.dumps(out, indent=1))
