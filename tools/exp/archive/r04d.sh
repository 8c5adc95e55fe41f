#!/bin/bash
# round 4, GPU pass D: sub-pixel upsample convs: tests, A/B, per-op profiles of three configs
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04d
mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv8p_gpu.py -x -q -rf > $OUT/tests.log 2>&1
tail -12 $OUT/tests.log
timeout 900 python tools/conv8p_bench.py --only up_ --bns 0 --out $OUT/subpixel_bench.json > $OUT/subpixel_bench.txt 2>&1
cat $OUT/subpixel_bench.txt | cut -c1-220
timeout 300 python tools/op_profile.py --batch 16 --latent 64 --model wukong --top 40 > $OUT/opprof_wukong_b16.txt 2>&1
timeout 300 python tools/op_profile.py --batch 8 --latent 96 --model sd2 --top 40 > $OUT/opprof_sd768_b8.txt 2>&1
timeout 300 python tools/op_profile.py --batch 2 --latent 64 --model sd2 --top 60 > $OUT/opprof_sd2_b2.txt 2>&1
head -7 $OUT/opprof_wukong_b16.txt $OUT/opprof_sd768_b8.txt $OUT/opprof_sd2_b2.txt
