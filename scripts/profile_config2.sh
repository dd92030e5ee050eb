#!/bin/bash
# GPU box: configs[1] (1 k landmarks, unknown correspondence) in its three association modes + rocprofv3 kernel stats of the waited one.
# Usage: scripts/profile_config2.sh <tag>
set -e -o pipefail
TAG=${1:-c2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/scripts/bench_config2.py --check > $OUT/device.json 2>/dev/null
python3 $REPO/scripts/bench_config2.py --verified --check > $OUT/verified.json 2>/dev/null
python3 $REPO/scripts/bench_config2.py --host-decision > $OUT/host.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/scripts/bench_config2.py > $OUT/under_rocprof.json 2> $OUT/rocprof.err
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
for f in device verified host; do python3 -c "
import json,sys
b=json.loads(open(sys.argv[1]).readline()); print(sys.argv[1].split('/')[-1], round(b['value']), b.get('slam_iterations_per_s'), b.get('parity'))" $OUT/$f.json; done
head -6 $OUT/kernel_stats.csv | cut -c1-200
