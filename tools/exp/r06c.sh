#!/bin/bash
# round 6: lean dense kernel (csrc/dense.hip) -- bit-identity tests, the kernel test file, then same-box A/B of one UNet evaluation
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "lean_dense or caller_owned" > gpurun_out/r06c/pytest_lean.log 2>&1; tail -15 gpurun_out/r06c/pytest_lean.log
timeout 600 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --arms "generic:gemm_lean_dense=0" "lean:gemm_lean_dense=1" --rounds 7 --out gpurun_out/r06c/ab_sd2_b2.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c/ab_sd2_b2.txt
timeout 600 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --arms "generic:gemm_lean_dense=0" "lean:gemm_lean_dense=1" --rounds 5 --iters 5 --out gpurun_out/r06c/ab_wk_b16.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c/ab_wk_b16.txt
timeout 600 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --arms "generic:gemm_lean_dense=0" "lean:gemm_lean_dense=1" --rounds 5 --iters 5 --out gpurun_out/r06c/ab_sd2_768.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c/ab_sd2_768.txt
