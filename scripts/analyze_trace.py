#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV: per-kernel average duration, and for the last `tail` dispatches the
start-to-start period and the idle gap before each kernel.  Usage: analyze_trace.py <kernel_trace.csv> [tail]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9]+)(<[^(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:50]


dur = defaultdict(list)
gap = defaultdict(list)
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    dur[k].append(e - s)
    if prev_end is not None:
        gap[k].append(s - prev_end)
    prev_end = e
print("%-58s %7s %9s %9s" % ("kernel", "calls", "avg us", "gap-before us"))
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    g = sorted(gap[k])
    print("%-58s %7d %9.2f %9.2f (median)" % (k[:58], len(dur[k]), sum(dur[k]) / len(dur[k]) / 1e3, (g[len(g) // 2] if g else 0) / 1e3))
print("\nlast %d dispatches: start-to-start us | duration us | gap before us | kernel" % tail)
last = rows[-tail:]
for i, r in enumerate(last):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    p = int(last[i - 1]["Start_Timestamp"]) if i else s
    pe = int(last[i - 1]["End_Timestamp"]) if i else s
    print("%8.2f %8.2f %8.2f  %s" % ((s - p) / 1e3, (e - s) / 1e3, (s - pe) / 1e3, short(r["Kernel_Name"])[:70]))
