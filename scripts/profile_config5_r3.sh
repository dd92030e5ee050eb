#!/bin/bash
# GPU box: configs[4] shape on one GPU, F32 tile storage (profiles/round3_config5_1gpu.json)
set -e -o pipefail
OUT=gpurun_out/${1:-r3c5}
mkdir -p $OUT
for r in 0; do
  python scripts/bench_config5.py --batch 1 --steps 96 --warmup 16 > $OUT/c5_40k_b1_r$r.json 2>/dev/null
  python scripts/bench_config5.py --batch 12 --steps 384 > $OUT/c5_40k_b12_r$r.json 2>/dev/null
  python scripts/bench_config5.py --batch 32 --steps 512 > $OUT/c5_40k_b32_r$r.json 2>/dev/null
done
python scripts/bench_config5.py --landmarks 49400 --batch 12 --steps 384 > $OUT/c5_50k_b12_r0.json 2>/dev/null
python scripts/bench_config5.py --landmarks 49400 --batch 1 --steps 64 --warmup 16 > $OUT/c5_50k_b1_r0.json 2>/dev/null
for f in $OUT/c5_*.json; do python -c "
import json,sys
b=json.loads(open(sys.argv[1]).readline()); r=b['roofline']
print(sys.argv[1].split('/')[-1], round(b['value']), round(r['frac'],4), r['kernel'], round(r['avg_launch_ms'],3))" $f; done
