#!/bin/bash
# What made round 3's three-counter run (TCC_HIT_sum TCC_MISS_sum FETCH_SIZE on ~1 700 dispatches) outlive the runner's limit?
# (a) the same three counters, pass kernel only; (b) the same three counters on EVERY dispatch of a short run, bounded by timeout.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_overhead
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/scripts/bench_config5.py --landmarks 40000 --steps 192 --warmup 64 --batch 64 --storage f32_mixed"
t0=$(date +%s.%N)
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum FETCH_SIZE --kernel-include-regex 'k_flush_strip32' --output-format csv -d $OUT/a -- $CMD > $OUT/a.json 2> $OUT/a.err; ra=$?
t1=$(date +%s.%N)
echo "(a) three counters, pass kernel only (4 dispatches): exit $ra, $(echo "$t1 - $t0" | bc) s"
timeout -k 10 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum FETCH_SIZE --output-format csv -d $OUT/b -- $CMD > $OUT/b.json 2> $OUT/b.err; rb=$?
t2=$(date +%s.%N)
echo "(b) three counters, every dispatch (~520): exit $rb, $(echo "$t2 - $t1" | bc) s"
grep -h '"value"' $OUT/a.json $OUT/b.json | cut -c1-160
f=$(find $OUT/b -name '*counter_collection.csv' | head -1); [ -n "$f" ] && echo "(b) rows: $(wc -l < $f)"
t3=$(date +%s.%N)
timeout -k 10 100 $CMD > $OUT/c.json 2> $OUT/c.err
t4=$(date +%s.%N)
echo "(c) no profiler: $(echo "$t4 - $t3" | bc) s"
