#!/bin/bash
# GPU box: A/B of the alternating pass direction (cfg.pass_direction = 1 / 2), interleaved rounds, several map sizes.
# Usage: scripts/ab_pass_direction.sh <tag>
set -e -o pipefail
TAG=${1:-alt}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for round in 1 2 3; do
  for L in 10000 7000 5000 3536; do
    for B in 1 32; do
      for ALT in 0 1; do
        timeout -k 10 120 python3 $REPO/scripts/time_flush.py --landmarks $L --batch $B --batches $([ $B = 1 ] && echo 256 || echo 12) --pass-direction $((ALT + 1)) --label "alt=$ALT" 2>/dev/null | grep '^{' >> $OUT/ab.jsonl
      done
    done
  done
  echo "round $round done"
done
python3 - <<PY
import json, collections
rows = collections.defaultdict(list)
for l in open("$OUT/ab.jsonl"):
    r = json.loads(l)
    rows[(r["landmarks"], r["batch"], r["label"])].append((r["flush_ms"], r["steps_per_s"]))
for k in sorted(rows):
    print(k, " ".join("%.4f/%d" % v for v in rows[k]))
PY
