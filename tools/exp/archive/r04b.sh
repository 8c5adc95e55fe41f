#!/bin/bash
# round 4, GPU pass B: conv8p with tail split / PH forms: tests, A/B, per-op profiles
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv8p_gpu.py -x -q -rf > $OUT/conv8p_tests.log 2>&1
tail -15 $OUT/conv8p_tests.log
timeout 900 python tools/conv8p_bench.py --bns 0,128 --extra 9:160,8:96 --out $OUT/conv8p_bench.json > $OUT/conv8p_bench.txt 2>&1
cat $OUT/conv8p_bench.txt | cut -c1-330
timeout 300 python tools/op_profile.py --batch 16 --latent 64 --model wukong --top 80 > $OUT/opprof_wukong_b16_c8.txt 2>&1
timeout 300 python tools/op_profile.py --batch 8 --latent 96 --model sd2 --top 80 > $OUT/opprof_sd768_b8_c8.txt 2>&1
head -8 $OUT/opprof_wukong_b16_c8.txt $OUT/opprof_sd768_b8_c8.txt
