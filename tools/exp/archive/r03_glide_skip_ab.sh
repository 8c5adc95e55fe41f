#!/bin/bash
# GLIDE planner with the ResBlock skip 1x1 fused into conv2 (MDX_UNET_SKIP_FUSE=1, default) vs separate launches (=0)
cd /root/repo
timeout 600 python -m pytest tests/test_glide_gpu.py tests/test_configs_gpu.py -m gpu -q -k "glide" 2>&1 | tail -3
for rep in 1 2; do
  for sf in 0 1; do
    MDX_UNET_SKIP_FUSE=$sf timeout 300 python bench.py --config glide_256 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
f = d['roofline']['families']
print('skip_fuse $sf', 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'gemm_ms', f['gemm']['ms'], 'launches', f['gemm']['launches'])"
  done
done
