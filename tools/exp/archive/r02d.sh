#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02d; mkdir -p $OUT
bash tools/gpu_pass.sh r02d tests bench
echo "== op profiles"
timeout 200 python tools/op_profile.py --batch 16 --top 60 > $OUT/op_profile_b16.txt 2>&1; head -70 $OUT/op_profile_b16.txt
timeout 200 python tools/op_profile.py --batch 2 --top 400 > $OUT/op_profile_b2.txt 2>&1; head -8 $OUT/op_profile_b2.txt
bash tools/gpu_pass.sh r02d configs
