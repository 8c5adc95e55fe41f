#!/bin/bash
# non-temporal weight DMA for launches whose weight tiles are read by exactly one block (one row of M tiles).  Record of a finished
# experiment: the MDX_GEMM_W_NT knob was removed again (slower, profiles/r02_o_weight_stream.txt)
export PYTHONPATH=.
mkdir -p gpurun_out/r02p
MDX_GEMM_W_NT=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv or split" 2>&1 | grep -v amdgpu.ids | tail -2
for nt in 0 1 0 1; do
  echo "-- MDX_GEMM_W_NT=$nt"
  MDX_GEMM_W_NT=$nt python tools/gemm_bench.py --batches 2 --iters 100 --only conv8,proj8 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02p/nt.txt
for nt in 0 1 0 1; do
  MDX_GEMM_W_NT=$nt python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt=$nt', r['value'], r['per_unet_step_ms'])"
done | tee -a gpurun_out/r02p/nt.txt
