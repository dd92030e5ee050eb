#!/bin/bash
# A/B two builds of libekfslam on the SAME box, interleaved rounds (cdna guide rule 24).
for r in 1 2 3; do
  for v in A B; do
    export EKF_LIB_PATH=$PWD/build_ab/lib$v.so
    h=$(python bench.py --no-cpu-baseline --no-other-configs --steps 1024 --warmup 64 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.readline()); print(round(b['value']))")
    c=$(python scripts/bench_config2.py 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.readline()); print(round(b['value']))")
    echo "round $r lib$v: 10k as-written $h steps/s | config2 $c update-steps/s"
  done
done
