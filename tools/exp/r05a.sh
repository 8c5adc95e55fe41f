#!/bin/bash
# round 5, GPU pass A: new parity tests (dense GroupNorm fold, DiffusionWrapper keys, full-size inpainting) + evaluation A/Bs
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05a
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_configs_gpu.py -m gpu -q -x \
    -k "dense_with_fused or wrapper_keys or proj_in or tiny_unet_forward or inpaint_wukong_full or conv3x3_with_fused_input or tiny_inpaint" \
    > $OUT/pytest_new.log 2>&1
tail -15 $OUT/pytest_new.log
timeout 500 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --rounds 5 --iters 20 --out $OUT/ab_sd2_b2.json \
    --arms "base:unet_gn_proj_fuse=0" "pf1024:unet_gn_proj_fuse=1024" "pf64:unet_gn_proj_fuse=64" \
           "fix8:unet_gn_proj_fuse=0,gemm_splitk_fixup_max=8" "pf1024fix8:unet_gn_proj_fuse=1024,gemm_splitk_fixup_max=8" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_sd2_b2.txt
timeout 400 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --rounds 3 --iters 5 --out $OUT/ab_wukong_b16.json \
    --arms "base:unet_gn_proj_fuse=0" "pf1024:unet_gn_proj_fuse=1024" "pf256:unet_gn_proj_fuse=256" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_wukong_b16.txt
timeout 400 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --rounds 3 --iters 5 --out $OUT/ab_sd2_b8_l96.json \
    --arms "base:unet_gn_proj_fuse=0" "pf1024:unet_gn_proj_fuse=1024" "pf256:unet_gn_proj_fuse=256" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_sd2_b8_l96.txt
timeout 300 python bench.py --no-other-configs --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json | cut -c1-600
