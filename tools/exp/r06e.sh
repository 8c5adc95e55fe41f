#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06e
timeout 600 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --arms "generic:gemm_lean_dense=0" "lean:gemm_lean_dense=1" "lean_nopre:gemm_lean_dense=2" --rounds 9 --out gpurun_out/r06e/ab_sd2_b2.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06e/ab_sd2_b2.txt
for lean in 0 1 2; do
  timeout 200 python tools/gemm_trace.py --warm --lean $lean --only proj16,proj32 2>&1 | grep -v amdgpu.ids > gpurun_out/r06e/gemm_trace_warm_lean$lean.txt
  grep -E "^#|per-block" gpurun_out/r06e/gemm_trace_warm_lean$lean.txt
  timeout 200 python tools/gemm_trace.py --lean $lean --only proj16,proj32 2>&1 | grep -v amdgpu.ids > gpurun_out/r06e/gemm_trace_cold_lean$lean.txt
  grep -E "per-block" gpurun_out/r06e/gemm_trace_cold_lean$lean.txt
done
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "groupnorm or splitk or gn_" > gpurun_out/r06e/pytest_gn.log 2>&1; tail -5 gpurun_out/r06e/pytest_gn.log
timeout 300 python tools/op_profile.py --batch 2 --top 400 2>&1 | grep -v amdgpu.ids > gpurun_out/r06e/op_profile_b2.txt; head -6 gpurun_out/r06e/op_profile_b2.txt; grep groupnorm gpurun_out/r06e/op_profile_b2.txt | sort -k3 -n -r | head -30
