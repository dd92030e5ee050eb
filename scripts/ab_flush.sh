#!/bin/bash
# A/B the pass over P on ONE box, interleaved rounds.  Variants: "label|lib (relative to repo, or - for the product build)|ENV=.. ENV=.."
# Usage: scripts/ab_flush.sh <out.log> <rounds> <extra time_flush args> -- variant...
OUT=$1; ROUNDS=$2; shift 2
ARGS=()
while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
mkdir -p $(dirname $OUT)
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    IFS='|' read -r label lib envs <<< "$v"
    (
      [ "$lib" != "-" ] && export EKF_LIB_PATH=$PWD/$lib
      for kv in $envs; do export $kv; done
      timeout -k 10 300 python scripts/time_flush.py --label "$label" "${ARGS[@]}" 2>/dev/null | grep '^{' >> $OUT
    ) || { echo "variant $label failed"; exit 1; }
  done
done
python - "$OUT" <<'PY'
import json, sys, collections
acc = collections.OrderedDict()
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    acc.setdefault(r["label"], []).append(r)
for k, rs in acc.items():
    f = [r["flush_ms"] for r in rs]; s = [r["steps_per_s"] for r in rs]; g = [r["gather_us"] for r in rs]
    print("%-34s flush ms %s  frac %.3f | gather us %s | steps/s %s | %s" % (k, " ".join("%.4f" % v for v in f), rs[-1]["frac"] if False else sum(r["frac"] for r in rs) / len(rs),
          " ".join("%.2f" % v for v in g), " ".join(str(v) for v in s), rs[0]["kernel"]))
PY
