#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06f
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "lean_dense" > gpurun_out/r06f/pytest_lean.log 2>&1; tail -4 gpurun_out/r06f/pytest_lean.log
timeout 900 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --arms "generic:gemm_lean_dense=0" "pf1:gemm_lean_dense=1" "pf0:gemm_lean_dense=2" "pf2:gemm_lean_dense=3" "pf3:gemm_lean_dense=4" --rounds 9 --out gpurun_out/r06f/ab_sd2_b2.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06f/ab_sd2_b2.txt
timeout 600 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --arms "generic:gemm_lean_dense=0" "pf1:gemm_lean_dense=1" "pf0:gemm_lean_dense=2" --rounds 5 --iters 5 --out gpurun_out/r06f/ab_wk_b16.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06f/ab_wk_b16.txt
timeout 600 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --arms "generic:gemm_lean_dense=0" "pf1:gemm_lean_dense=1" "pf0:gemm_lean_dense=2" --rounds 5 --iters 5 --out gpurun_out/r06f/ab_sd2_768.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06f/ab_sd2_768.txt
