#!/bin/bash
# round 6, pass AB: does the guidance-duplicate prefix pay at UNet batch 2 once the batch-1 level-0 shapes have measured tile rows?
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/tune
cp minddiffusion_amd/libmdx.so minddiffusion_amd/libmdx_old.so
T=gpurun_out/tune/gemm_tuned_b1.inc
cp minddiffusion_amd/csrc/gemm_tuned.inc $T
timeout 400 python tools/tune_gemm.py --model sd2 --batch 1 --latent 64 --only-m 4096 --merge --gain 0.03 --reps 9 --out $T --log gpurun_out/tune/dup_sd2_b1.log 2>&1 | grep -v amdgpu.ids | tail -25
cp $T minddiffusion_amd/csrc/gemm_tuned.inc

make -C minddiffusion_amd/csrc -j16 2>&1 | grep -E "error|Error"
OLD=$PWD/minddiffusion_amd/libmdx_old.so
for v in old new; do
  if [ $v = old ]; then L=$OLD; else L=$PWD/minddiffusion_amd/libmdx.so; fi
  echo "== $v"
  MDX_LIBRARY=$L timeout 200 python tools/eval_ab.py --guidance --model sd2 --batch 2 --latent 64 --rounds 7 --iters 30 --arms "plain:unet_cfg_dup=0" "dup:unet_cfg_dup=2" "plain2:unet_cfg_dup=0" "dup2:unet_cfg_dup=2" 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/tune/dup_b2_ab.txt
MDX_LIBRARY=$PWD/minddiffusion_amd/libmdx.so timeout 200 python tools/op_profile.py --model sd2 --batch 2 --latent 64 --guidance --top 24 2>&1 | grep -v amdgpu.ids | head -34 | tee gpurun_out/tune/dup_b2_opprof.txt
