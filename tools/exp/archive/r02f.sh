#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== correctness with 256-row generic tiles"
MDX_GEMM_BM=256 timeout 600 python -m pytest tests/test_unet_gpu.py -q -k "sd2_full_size_single_step or wukong_full" 2>&1 | tail -3
MDX_GEMM_BM=256 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or geglu or conv" 2>&1 | tail -3
echo "== B=16 token GEMMs: default vs 256-row tiles"
for v in "default:X=1" "BM256:MDX_GEMM_BM=256"; do
  name=${v%%:*}; envs=${v#*:}
  echo "-- $name"
  env $envs timeout 300 python tools/gemm_bench.py --batches 16 --only geglu,ff2,qk,proj --iters 20 2>&1 | grep "B=16"
done
