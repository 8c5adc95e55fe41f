#!/bin/bash
# A/B of the column-block width of the column-statistics GroupNorm (option gn_col_chunks: 16-byte chunks per pixel row).
cd /root/repo
for rep in 1 2; do
  for cc in 4 8 16 32; do
    MDX_GN_COL_CHUNKS=$cc timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
f = d['roofline']['families']
print('chunks $cc', 'value', d['value'], 'ms_per_step/50', round(d['ms_per_step'] / 50, 4), 'groupnorm_ms', f['groupnorm']['ms'])"
  done
done
MDX_GN_COL_CHUNKS=16 timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "groupnorm or colstats" 2>&1 | tail -2
