#!/bin/bash
# round 5, GPU pass C: the "lone block" K-loop rework (fragment reads ahead of the DMA issue + sched_barrier pins, dense issue through
# the scalar offset), GroupNorm / LayerNorm-fold parameter prefetch -- parity gates, same-box A/B against the round-4 library
# (minddiffusion_amd/libmdx_base.so = HEAD before this work), phase traces, GLIDE per-shape profile, VALU probe.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05c
mkdir -p $OUT
BASE=$PWD/minddiffusion_amd/libmdx_base.so
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }

timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --durations=6 > $OUT/pytest_kernels.log 2>&1
tail -12 $OUT/pytest_kernels.log; stamp kernel tests
timeout 300 python -m pytest tests/test_unet_gpu.py -m gpu -q -x --durations=6 \
    -k "tiny_unet_forward or constructor_variants or sd2_full_size_single_step or wukong_full_size_single_step or sd2_768_single_step" > $OUT/pytest_unet.log 2>&1
tail -12 $OUT/pytest_unet.log; stamp unet tests

for rep in 1; do
  MDX_LIBRARY=$BASE timeout 200 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --rounds 5 --iters 20 --arms "r4lib:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_sd2_b2.txt
  timeout 300 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --rounds 5 --iters 20 \
      --arms "new:" "dense0:gemm_dense_issue=0" "gnpre0:gn_prefetch=0" "lnpre0:gemm_ln_prefetch=0" "allopt0:gemm_dense_issue=0,gn_prefetch=0,gemm_ln_prefetch=0" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_sd2_b2.txt
done
stamp sd2 b2 ab
MDX_LIBRARY=$BASE timeout 200 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --rounds 3 --iters 5 --arms "r4lib:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_big.txt
timeout 200 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --rounds 3 --iters 5 --arms "new:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_big.txt
MDX_LIBRARY=$BASE timeout 200 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --rounds 3 --iters 5 --arms "r4lib:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_big.txt
timeout 200 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --rounds 3 --iters 5 --arms "new:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_big.txt
stamp big ab

for mode in "" "--warm"; do
  timeout 200 python tools/gemm_trace.py --only proj16_1280,proj32_640,geglu32_640,conv32_640_640,conv64_320_320 $mode 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_trace.txt
done
stamp trace
timeout 120 python tools/exp/r05_valu_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/valu_probe.txt
timeout 240 python tools/glide_op_profile.py --out $OUT/glide_ops.json 2>&1 | grep -v amdgpu.ids | tee $OUT/glide_ops.txt
stamp glide profile
timeout 400 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
stamp bench
timeout 240 python tools/shape_sweep.py --model sd2 --latents 32,64 --batches 1-5 2>&1 | grep -v amdgpu.ids | tee $OUT/shape_sweep.txt
stamp sweep
