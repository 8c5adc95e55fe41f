#!/bin/bash
# round 6, pass T: the global options re-measured on the round-6 kernels (whole evaluation, hipGraph replay, arms interleaved in one process)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06u
mkdir -p $OUT
timeout 600 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --rounds 5 --iters 20 --arms "base:" "fix6:gemm_splitk_fixup_max=6" "fix8:gemm_splitk_fixup_max=8" "fix2:gemm_splitk_fixup_max=2" "gnsk0:unet_gn_splitk_fuse=0" "gnsk1024:unet_gn_splitk_fuse=1024" "gnconv:unet_gn_conv_fuse=4096" "gnproj:unet_gn_proj_fuse=1" "pf0:gemm_lean_dense=2" "pf2:gemm_lean_dense=3" "stream512:unet_conv_stream=512" "stream0:unet_conv_stream=0" "base2:" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_b2.txt
timeout 600 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --rounds 3 --iters 10 --arms "base:" "fix6:gemm_splitk_fixup_max=6" "gnconv:unet_gn_conv_fuse=4096" "pf0:gemm_lean_dense=2" "pf2:gemm_lean_dense=3" "base2:" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_b8.txt
