#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
timeout 300 python tools/exp/r06_forms.py sd2 2 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r06d/forms_sd2_b2.txt; tail -40 gpurun_out/r06d/forms_sd2_b2.txt
for lean in 0 1; do
  timeout 200 python tools/gemm_trace.py --lean $lean --only proj16,proj32,qk64,ff2_16,geglu32 2>&1 | grep -v amdgpu.ids > gpurun_out/r06d/gemm_trace_lean$lean.txt
  grep -E "^#|^[a-z].*kernel span|per-block" gpurun_out/r06d/gemm_trace_lean$lean.txt
done
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "lean_dense" > gpurun_out/r06d/pytest_lean.log 2>&1; tail -8 gpurun_out/r06d/pytest_lean.log
