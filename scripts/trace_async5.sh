#!/bin/bash
# GPU box: kernel timeline of configs[4] with cfg.async_flush -- do the corrections of batch b+1 run WHILE pass b does?
# Prints, per pass launch, how many gather launches started inside its interval and the gathers' durations inside / outside.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_async5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $REPO/scripts/bench_config5.py --landmarks 40000 --steps 384 --warmup 64 --batch 64 --storage ${STORAGE:-f32_split} --async-flush > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
passes = [(s, e) for s, e, k in ev if "k_flush" in k]
gath = [(s, e) for s, e, k in ev if "k_gather" in k]
print("passes", len(passes), "gathers", len(gath))
for ps, pe in passes[-5:]:
    inside = [(s, e) for s, e in gath if ps <= s < pe]
    print("pass %.2f ms: %d gathers started inside, their mean duration %.1f us" % ((pe - ps) / 1e6, len(inside), sum(e - s for s, e in inside) / max(len(inside), 1) / 1e3))
outside = [(s, e) for s, e in gath if not any(ps <= s < pe for ps, pe in passes)]
print("gathers outside any pass: %d, mean %.1f us" % (len(outside), sum(e - s for s, e in outside) / max(len(outside), 1) / 1e3))
t0 = passes[-4][0]
for s, e, k in ev:
    if t0 - 200e3 <= s <= t0 + 400e3:
        print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, k.replace("(anonymous namespace)::", "").replace("void ", "")[:60]))
PY
