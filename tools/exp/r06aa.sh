#!/bin/bash
# round 6, pass AA: tile table rows for the half-batch prefix of the guidance-duplicate evaluation (level-0 shapes at batch 8 / 64^2 and batch 4 / 96^2)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/tune
cp minddiffusion_amd/libmdx.so minddiffusion_amd/libmdx_old.so
T=gpurun_out/tune/gemm_tuned_dup.inc
cp minddiffusion_amd/csrc/gemm_tuned.inc $T
timeout 300 python tools/tune_gemm.py --model wukong --batch 8 --latent 64 --only-m 32768 --merge --gain 0.03 --reps 7 --out $T --log gpurun_out/tune/dup_wukong_b8.log 2>&1 | grep "KEEP\|entries"
timeout 300 python tools/tune_gemm.py --model sd2 --batch 4 --latent 96 --only-m 36864 --merge --gain 0.03 --reps 7 --out $T --log gpurun_out/tune/dup_sd2_b4_l96.log 2>&1 | grep "KEEP\|entries"
cp $T minddiffusion_amd/csrc/gemm_tuned.inc
make -C minddiffusion_amd/csrc -j16 2>&1 | grep -E "error|Error"
OLD=$PWD/minddiffusion_amd/libmdx_old.so
for v in old new old new; do
  if [ $v = old ]; then L=$OLD; else L=$PWD/minddiffusion_amd/libmdx.so; fi
  echo "== $v"
  MDX_LIBRARY=$L timeout 200 python tools/eval_ab.py --guidance --model wukong --batch 16 --latent 64 --rounds 5 --iters 10 --arms "dup:unet_cfg_dup=2" 2>&1 | grep -v amdgpu.ids
  MDX_LIBRARY=$L timeout 200 python tools/eval_ab.py --guidance --model sd2 --batch 8 --latent 96 --rounds 5 --iters 10 --arms "dup:unet_cfg_dup=2" 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/tune/dup_ab.txt
