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


def avg_counter(kind, counter, kernel_substr):
    f = glob.glob(os.path.join(src, "pmc_%s" % kind, "*", "*_counter_collection.csv"))[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


bench = json.load(open(os.path.join(src, "bench.json")))
fetch_kib, nf = avg_counter("fetch", "FETCH_SIZE", "k_downdate")
write_kib, nw = avg_counter("write", "WRITE_SIZE", "k_downdate")
rec = {
    "kernel": "k_downdate",
    "landmarks": bench["config"]["landmarks"],
    "tile": bench["config"]["tile"],
    "FETCH_SIZE_KiB_avg": fetch_kib, "fetch_dispatches": nf,
    "WRITE_SIZE_KiB_avg": write_kib, "write_dispatches": nw,
    "correction": "gfx950: FETCH_SIZE x2 (128-B requests tallied at 64 B, 16-B/lane streaming reads); WRITE_SIZE exact",
    "hbm_read_bytes_per_launch": 2.0 * fetch_kib * 1024.0,
    "hbm_write_bytes_per_launch": write_kib * 1024.0,
    "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py, %s" % tag,
}
rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
json.dump(rec, open(os.path.join(dst, "downdate_pmc.json"), "w"), indent=1)
shutil.copy(os.path.join(dst, "downdate_pmc.json"), os.path.join(dst, "%s_downdate_pmc.json" % rnd))
print(json.dumps(rec, indent=1))
