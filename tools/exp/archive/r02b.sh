#!/bin/bash
# round-2 experiment pass b: HALO8 + rolled HALO schedule
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02b; mkdir -p $OUT
echo "== correctness: default build"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "conv or halo or gemm" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_unet_gpu.py -q -k "tiny or zero or wukong_style" 2>&1 | tail -3
echo "== correctness: rolled schedule (NSB=4), 128- and 256-pixel patches"
MDX_HALO_NSB=4 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "conv or halo" 2>&1 | tail -3
MDX_HALO_NSB=4 MDX_GEMM_BM=256 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "conv or halo" 2>&1 | tail -3
echo "== B=16 convs: tile / schedule variants"
for v in "BM128_NSB2:MDX_GEMM_BM=128" "BM128_NSB4:MDX_GEMM_BM=128 MDX_HALO_NSB=4" "BM256_NSB3:MDX_GEMM_BM=256" "BM256_NSB4:MDX_GEMM_BM=256 MDX_HALO_NSB=4"; do
  name=${v%%:*}; envs=${v#*:}
  echo "-- $name ($envs)"
  env $envs timeout 300 python tools/gemm_bench.py --batches 16 --only conv64,conv32,conv16 --iters 20 2>&1 | grep "B=16"
done
echo "== B=2 convs at 8x8: HALO8 on/off"
MDX_GEMM_HALO8=1 timeout 200 python tools/gemm_bench.py --batches 2 --only conv8 --iters 30 --splits 0,5,8,10,13,16,20 2>&1 | grep "B= 2"
echo "-- off"
MDX_GEMM_HALO8=0 timeout 200 python tools/gemm_bench.py --batches 2 --only conv8 --iters 30 --splits 0,10,13 2>&1 | grep "B= 2"
echo "== bench A/B (per_unet_step_ms)"
for v in "halo8_on:MDX_GEMM_HALO8=1" "halo8_off:MDX_GEMM_HALO8=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_$name.json 2>$OUT/bench_$name.err
  python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print('$name', d['value'], d['per_unet_step_ms'], d['roofline']['families']['gemm'])"
done
