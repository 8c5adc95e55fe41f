#!/bin/bash
# round 6, pass Z: guidance-duplicate prefix (UNetModel._dup_body) -- parity and same-box A/B
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06z
mkdir -p $OUT
timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -x -q > $OUT/pytest_u.log 2>&1; tail -5 $OUT/pytest_u.log
ab() { timeout 400 python tools/eval_ab.py --guidance "$@" --arms "plain:unet_cfg_dup=0" "dup:unet_cfg_dup=2" "plain2:unet_cfg_dup=0" "dup2:unet_cfg_dup=2" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt; }
ab --model sd2 --batch 8 --latent 96 --rounds 5 --iters 10
ab --model wukong --batch 16 --latent 64 --rounds 5 --iters 10


