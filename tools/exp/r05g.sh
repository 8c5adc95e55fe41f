#!/bin/bash
# round 5, GPU pass G: all four benchmarked configurations, round-4 library vs the adopted build + re-tuned table (alternating, one box)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05g
mkdir -p $OUT
OLD=$PWD/minddiffusion_amd/libmdx_base.so
for c in sd2_512 wukong_512_plms sd2_768 glide_256; do
  for v in old new old new; do
    if [ $v = old ]; then L=$OLD; else L=$PWD/minddiffusion_amd/libmdx.so; fi
    MDX_LIBRARY=$L timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', r['value'], r.get('per_unet_step_ms'))" | tee -a $OUT/bench_ab.txt
  done
done
