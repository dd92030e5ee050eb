#!/usr/bin/env python3
"""One-screen summary of a bench.py JSON line: python scripts/show_bench.py file.json"""
import json
import sys
b = json.loads(open(sys.argv[1]).readline())
r = b["roofline"]
print("headline  %8.0f steps/s  %.4f ms/step  frac %.4f  %s  launch %.4f ms  traffic %s" % (b["value"], b["ms_per_step"], r["frac"], r["kernel"], r["avg_launch_ms"], r["traffic"]))
for k in ("deferred", "deferred_b32", "deferred_lookahead"):
    if k in b:
        d = b[k]; r = d["roofline"]
        print("%-9s %8.0f steps/s  %.4f ms/step  frac %.4f  %s  launch %.4f ms  eff %.0f GB/s" % (k[:9], d["value"], d["ms_per_step"], r["frac"], r["kernel"], r["avg_launch_ms"], d["effective_GBps"]))
if "cpu_baseline" in b:
    print("cpu       %8.1f %s on %d cores (%s)" % (b["cpu_baseline"]["value"], b["cpu_baseline"]["unit"], b["cpu_baseline"]["cores"], b["cpu_baseline"]["kind"]))
for k, c in b.get("other_configs", {}).items():
    if "error" in c:
        print("%-58s ERROR %s" % (k, str(c["error"])[:80]))
        continue
    r = c.get("roofline") or {}
    print("%-58s %8.0f steps/s  %s" % (k, c["value"], ("%s  launch %.3f ms  bound %s %.3f" % (r["kernel"], r["avg_launch_ms"], r["bound"], r["frac"])) if r else ""))
print("n_gpus", b["n_gpus"], "transport", b["config"]["transport"], "digest", b["config"]["state_digest"])
