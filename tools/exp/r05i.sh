#!/bin/bash
# round 5, GPU pass I: attention full-tile DMA issue through the scalar offset (option attn_fast_stage) + kernel-argument warm-up build
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05i
mkdir -p $OUT
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_text_encoder_gpu.py -m gpu -q -x -k "attention or attn or encoder" > $OUT/pytest_attn.log 2>&1; tail -2 $OUT/pytest_attn.log
for fs in 0 1 0 1; do
  echo "attn_fast_stage=$fs" | tee -a $OUT/attn_bench.txt
  MDX_ATTN_FAST_STAGE=$fs timeout 200 python tools/attn_bench.py --forms o3,o3s --rounds 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/attn_bench.txt
done
timeout 300 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --rounds 3 --iters 5 --arms "fs1:" "fs0:attn_fast_stage=0" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
timeout 300 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --rounds 3 --iters 5 --arms "fs1:" "fs0:attn_fast_stage=0" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
timeout 300 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --rounds 5 --iters 20 --arms "fs1:" "fs0:attn_fast_stage=0" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
for lib in new kt new kt; do
  path=$PWD/minddiffusion_amd/libmdx_$lib.so; [ $lib = new ] && path=$PWD/minddiffusion_amd/libmdx.so
  MDX_LIBRARY=$path timeout 200 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --rounds 5 --iters 20 --arms "$lib:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_kt.txt
done
