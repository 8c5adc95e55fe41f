#!/bin/bash
# A/B of the GroupNorm column-partial fold depth (norm.hip gn_apply_kernel): libmdx_old.so = 4 loads in flight, libmdx.so = 16.
cd /root/repo
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export MDX_LIBRARY=/root/repo/minddiffusion_amd/libmdx_old.so; else unset MDX_LIBRARY; fi
    timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
f = d['roofline']['families']
print('$lib', 'value', d['value'], 'ms_per_step/50', round(d['ms_per_step'] / 50, 4), 'groupnorm_ms', f['groupnorm']['ms'], 'gemm_ms', f['gemm']['ms'])"
  done
done
unset MDX_LIBRARY
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "groupnorm or colstats" 2>&1 | tail -2
