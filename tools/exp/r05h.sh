#!/bin/bash
# round 5, GPU pass H: four-stage weight ring in the HALO conv (option halo_nsb = 4 forces it; 0 = the rule: 3 on 64-column tiles)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05h
mkdir -p $OUT
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1; tail -2 $OUT/pytest_kernels.log
MDX_HALO_NSB=4 timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv" > $OUT/pytest_nsb4.log 2>&1; tail -2 $OUT/pytest_nsb4.log
timeout 300 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --rounds 5 --iters 20 --arms "nsb_auto:" "nsb4:halo_nsb=4" "nsb3:halo_nsb=3" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_nsb.txt
timeout 300 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --rounds 3 --iters 5 --arms "nsb_auto:" "nsb4:halo_nsb=4" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_nsb.txt
