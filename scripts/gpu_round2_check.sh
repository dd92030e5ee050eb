#!/bin/bash
# GPU box: the -m gpu suite, then the bench line, then the multi-rank rehearsal on the one GPU (gloo).  Usage: scripts/gpu_round2_check.sh <tag>
TAG=${1:-r2a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest.log 2>&1
rc=$?
tail -25 $OUT/pytest.log
if [ $rc -ge 124 ]; then echo "pytest timed out"; exit $rc; fi
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
rb=$?
cat $OUT/bench.json; tail -5 $OUT/bench.err
if [ $rb -ge 124 ]; then echo "bench timed out"; exit $rb; fi
timeout -k 10 500 bash scripts/rehearse_multirank.sh 2000 > $OUT/rehearse.log 2>&1
rr=$?
tail -8 $OUT/rehearse.log
echo "pytest rc=$rc bench rc=$rb rehearse rc=$rr"
[ $rc -eq 0 ] && [ $rb -eq 0 ] && [ $rr -eq 0 ]
