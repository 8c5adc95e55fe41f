#!/bin/bash
# round 4, GPU pass E: same-box whole-evaluation A/B of the round's launch forms
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04e
mkdir -p $OUT
R3="gemm_conv8p=0,gemm_dense8p=0,unet_subpixel_upsample=0"
for cfg in "wukong 16 64" "sd2 8 96" "sd2 2 64"; do
  set -- $cfg
  timeout 600 python tools/eval_ab.py --model $1 --batch $2 --latent $3 --out $OUT/ab_$1_b$2_l$3.json \
     --arms "round3:$R3" "conv8p:gemm_dense8p=0,unet_subpixel_upsample=0" "c8+sub:gemm_dense8p=0" "all:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
