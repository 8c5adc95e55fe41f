#!/bin/bash
# A/B of gemm_splitk_fixup_max (split-K launches of at most this many splits reduce in the kernel): 4 (default) vs 5, 6, 8.
cd /root/repo
for rep in 1 2; do
  for fm in 4 5 6 8; do
    MDX_GEMM_SPLITK_FIXUP_MAX=$fm timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
f = d['roofline']['families']
print('fixup_max $fm', 'value', d['value'], 'ms_per_step/50', round(d['ms_per_step'] / 50, 4), 'launches', sum(v.get('launches', 0) for v in f.values()), 'gemm_ms', f['gemm']['ms'])"
  done
done
