#!/bin/bash
# round 5, GPU pass B: remaining new tests, phase traces of the token GEMMs, GN fold A/B after the prologue change
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_configs_gpu.py tests/test_kernels_gpu.py -m gpu -q \
    -k "dense_with_fused or wrapper_keys or proj_in or inpaint_wukong_full or tiny_inpaint" > $OUT/pytest_new.log 2>&1
tail -8 $OUT/pytest_new.log
for mode in "" "--warm"; do
  timeout 200 python tools/gemm_trace.py --only proj32_640,proj16_1280,geglu32_640 $mode 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_trace.txt
done
timeout 200 python tools/gemm_trace.py --only proj16_1280 --split 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_trace.txt
timeout 500 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --rounds 5 --iters 20 --out $OUT/ab_sd2_b2.json \
    --arms "base:unet_gn_proj_fuse=0" "pf1024:unet_gn_proj_fuse=1024" "pf256:unet_gn_proj_fuse=256" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_sd2_b2.txt
timeout 400 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --rounds 3 --iters 5 --out $OUT/ab_wukong_b16.json \
    --arms "base:unet_gn_proj_fuse=0" "pf1024:unet_gn_proj_fuse=1024" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_wukong_b16.txt
