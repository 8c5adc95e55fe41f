#!/bin/bash
# retune the tile table with the round-2 kernels, one config after the other, merging into one file
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02g; mkdir -p $OUT
cp minddiffusion_amd/csrc/gemm_tuned.inc $OUT/gemm_tuned.inc
t0=$(date +%s)
timeout 400 python tools/tune_gemm.py --model sd2 --batch 2 --latent 64 --merge --out $OUT/gemm_tuned.inc --log $OUT/sd2_b2.log 2>&1 | tail -2
echo "sd2 b2 done $(( $(date +%s) - t0 )) s"
timeout 400 python tools/tune_gemm.py --model wukong --batch 16 --latent 64 --merge --out $OUT/gemm_tuned.inc --log $OUT/wukong_b16.log 2>&1 | tail -2
echo "wukong b16 done $(( $(date +%s) - t0 )) s"
timeout 400 python tools/tune_gemm.py --model sd2 --batch 8 --latent 96 --merge --out $OUT/gemm_tuned.inc --log $OUT/sd2_b8_l96.log 2>&1 | tail -2
echo "sd2 b8 l96 done $(( $(date +%s) - t0 )) s"
timeout 500 python tools/tune_gemm.py --model glide --merge --out $OUT/gemm_tuned.inc --log $OUT/glide.log 2>&1 | tail -2
echo "glide done $(( $(date +%s) - t0 )) s"
wc -l $OUT/gemm_tuned.inc
