#!/bin/bash
# usage: bench_variability.sh "28 32 36" 3 [batches per run, default 10]  -> batch, steps, update-steps/s, flush ms, roofline frac per run
for r in $(seq 1 ${2:-3}); do
  for b in $1; do
    timeout -k 10 200 python bench.py --batch $b --steps $((b*${3:-10})) --warmup $((b*4)) --no-cpu-baseline --no-other-configs 2>/dev/null > /tmp/bv.json || exit 1
    python - "$b" <<'PY'
import json, sys
d = json.load(open('/tmp/bv.json'))
print(sys.argv[1], d["steps"], round(d["value"]), round(d["roofline"]["avg_launch_ms"], 4), round(d["roofline"]["frac"], 3), flush=True)
PY
  done
done
