#!/bin/bash
# GPU box: the other configurations' numbers for profiles/<round>_config2.json / _config5_1gpu.json / _cpu_baselines.json.
# Usage: scripts/profile_configs.sh <tag>
set -e -o pipefail
TAG=${1:-r2q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
# configs[1]: the device-resident loop (cfg.device_assoc = 3, the default) is the primary; the other three modes beside it
python scripts/bench_config2.py --check > $OUT/config2_devloop.json 2>/dev/null
python scripts/bench_config2.py --batch 1 --check > $OUT/config2_devloop_b1.json 2>/dev/null
python scripts/bench_config2.py --batch 16 > $OUT/config2_devloop_b16.json 2>/dev/null
python scripts/bench_config2.py --verified --check > $OUT/config2_verified.json 2>/dev/null
python scripts/bench_config2.py --waited --check > $OUT/config2_waited.json 2>/dev/null
python scripts/bench_config2.py --host-decision --check > $OUT/config2_host.json 2>/dev/null
bash scripts/trace_config2.sh $TAG/c2trace --batch 8 > /dev/null
# configs[4] shape on one GPU: 40 k landmarks F32 tiles, streaming append; immediate, batch 12, batch 32; the full 50 k map
python scripts/bench_config5.py --batch 1 --steps 96 --warmup 16 > $OUT/config5_40k_b1.json 2>/dev/null
python scripts/bench_config5.py --batch 12 --steps 384 > $OUT/config5_40k_b12.json 2>/dev/null
python scripts/bench_config5.py --batch 32 --steps 512 > $OUT/config5_40k_b32.json 2>/dev/null
python scripts/bench_config5.py --landmarks 49400 --batch 12 --steps 384 > $OUT/config5_50k_b12.json 2>/dev/null
python scripts/bench_config5.py --landmarks 49400 --batch 1 --steps 64 --warmup 16 > $OUT/config5_50k_b1.json 2>/dev/null
# ... and the same shape with the pass in F32 arithmetic on the matrix pipe (cfg.pass_arith = EKF_ARITH_F32: "F32 mixed precision with F64 innovation solve")
python scripts/bench_config5.py --storage f32_mixed --batch 1 --steps 96 --warmup 16 > $OUT/config5_40k_mixed_b1.json 2>/dev/null
python scripts/bench_config5.py --storage f32_mixed --batch 12 --steps 384 > $OUT/config5_40k_mixed_b12.json 2>/dev/null
python scripts/bench_config5.py --storage f32_mixed --batch 32 --steps 512 > $OUT/config5_40k_mixed_b32.json 2>/dev/null
python scripts/bench_config5.py --storage f32_mixed --batch 64 --steps 512 > $OUT/config5_40k_mixed_b64.json 2>/dev/null
python scripts/bench_config5.py --storage f32_mixed --landmarks 49400 --batch 32 --steps 512 > $OUT/config5_50k_mixed_b32.json 2>/dev/null
python scripts/bench_cpu_restatements.py > $OUT/cpu_baselines.json 2>$OUT/cpu_baselines.err
for f in $OUT/config*.json; do python -c "
import json,sys
b=json.loads(open(sys.argv[1]).readline()); r=b.get('roofline',{})
print(sys.argv[1].split('/')[-1], round(b['value']), b.get('slam_iterations_per_s'), r.get('frac'), r.get('kernel'), r.get('avg_launch_ms'), b.get('parity'))" $f; done
