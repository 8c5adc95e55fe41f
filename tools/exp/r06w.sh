#!/bin/bash
# round 6, pass W: cross-attention as the epilogue of the query projection (mdx_gemm_desc.xattn_k) -- parity, UNet suites, whole-evaluation A/B
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06w
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "cross_attention_epilogue" > $OUT/pytest_k.log 2>&1; tail -5 $OUT/pytest_k.log
timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $OUT/pytest_u.log 2>&1; tail -4 $OUT/pytest_u.log
for cfg in "sd2 2 64" "sd2 8 96"; do
  set -- $cfg
  timeout 300 python tools/eval_ab.py --model $1 --batch $2 --latent $3 --rounds 7 --iters 20 --arms "base:unet_xattn_fuse=0" "fuse:unet_xattn_fuse=1" "base2:unet_xattn_fuse=0" "fuse2:unet_xattn_fuse=1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
