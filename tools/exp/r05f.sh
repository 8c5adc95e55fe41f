#!/bin/bash
# round 5, GPU pass F: parity gates on the adopted build (pinned + unrolled HALO taps, lean dense K loop), then an in-situ re-tune of
# the tile table on these kernels (all four benchmarked configurations); the table comes back under gpurun_out/tune/
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05f
mkdir -p $OUT gpurun_out/tune
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_conv8p_gpu.py tests/test_stchain_gpu.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1; tail -3 $OUT/pytest_kernels.log
T=gpurun_out/tune/gemm_tuned_r05.inc
cp minddiffusion_amd/csrc/gemm_tuned.inc $T
timeout 500 python tools/tune_gemm.py --model sd2 --batch 2 --latent 64 --merge --gain 0.03 --reps 9 --out $T --log gpurun_out/tune/r05_sd2_b2.log 2>&1 | grep "KEEP\|entries"
timeout 500 python tools/tune_gemm.py --model wukong --batch 16 --latent 64 --merge --gain 0.03 --reps 5 --out $T --log gpurun_out/tune/r05_wukong_b16.log 2>&1 | grep "KEEP\|entries"
timeout 500 python tools/tune_gemm.py --model sd2 --batch 8 --latent 96 --merge --gain 0.03 --reps 5 --out $T --log gpurun_out/tune/r05_sd2_b8_l96.log 2>&1 | grep "KEEP\|entries"
timeout 600 python tools/tune_gemm.py --model glide --merge --gain 0.03 --reps 5 --out $T --log gpurun_out/tune/r05_glide.log 2>&1 | grep "KEEP\|entries"
