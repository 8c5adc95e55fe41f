#!/bin/bash
# round 4, GPU pass C: gemm8p tests + A/B, conv8p picks re-check, per-op profiles
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04c
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gemm8p_gpu.py tests/test_conv8p_gpu.py -x -q -rf > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log
timeout 900 python tools/gemm8p_bench.py --out $OUT/gemm8p_bench.json > $OUT/gemm8p_bench.txt 2>&1
cat $OUT/gemm8p_bench.txt | cut -c1-200
timeout 300 python tools/op_profile.py --batch 16 --latent 64 --model wukong --top 60 > $OUT/opprof_wukong_b16.txt 2>&1
timeout 300 python tools/op_profile.py --batch 8 --latent 96 --model sd2 --top 60 > $OUT/opprof_sd768_b8.txt 2>&1
head -8 $OUT/opprof_wukong_b16.txt $OUT/opprof_sd768_b8.txt
