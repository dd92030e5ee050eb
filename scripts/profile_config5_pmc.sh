#!/bin/bash
# GPU box (through gpurun): hardware counters of the 64-pair pass of the configs[4] workload -- HBM bytes (FETCH_SIZE, WRITE_SIZE),
# matrix-pipe busy cycles, L2 hits / misses.  Usage: [STORAGE=f32_mixed|f32_split] scripts/profile_config5_pmc.sh <tag>  -> gpurun_out/<tag>/pmc_*/...
# The recipe (profiles/README.md "Counters"): one counter group per rocprofv3 run; the program itself after `--`; counters only for the
# pass kernel (--kernel-include-regex: every instrumented dispatch costs tens of milliseconds, and the workload has two small launches
# per update-step -- round 3's three-counter run on 800 steps instrumented 1 700 dispatches and ran into the runner's limit); a SHORT
# run (64 warm-up + 192 steps: four passes); a progress line between runs.
set -o pipefail
TAG=${1:-r04_c5pmc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STORAGE=${STORAGE:-f32_mixed}
CMD="python3 $REPO/scripts/bench_config5.py --landmarks 40000 --steps 192 --warmup 64 --batch 64 --storage $STORAGE"
RX='k_flush_strip32|k_flush_mfma32|k_flush_split3'
run() {   # name, counters...
  local name=$1; shift
  echo "[pmc] $name: $*"
  timeout -k 10 280 rocprofv3 --pmc "$@" --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc_$name -- $CMD > $OUT/$name.json 2> $OUT/$name.err \
    || { echo "[pmc] $name FAILED (exit $?)"; tail -3 $OUT/$name.err; return 1; }
  echo "[pmc] $name done: $(find $OUT/pmc_$name -name '*counter_collection.csv' | head -1)"
}
run fetch FETCH_SIZE && run write WRITE_SIZE && run mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE && run l2 TCC_HIT_sum TCC_MISS_sum
echo "[pmc] kernel trace"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.json 2> $OUT/stats.err
python3 $REPO/scripts/summarize_config5_pmc.py $OUT $OUT/config5_pmc.json $STORAGE
