#!/bin/bash
# ring depth of the generic kernel on the short-K token GEMMs at UNet batch 2: is the K loop latency-serialised?
export PYTHONPATH=.
mkdir -p gpurun_out/r02k
for cfg in default 64,2 64,3 64,4 64,5; do
  echo "-- MDX_GEMM_CFG=$cfg"
  if [ $cfg = default ]; then unset MDX_GEMM_CFG; else export MDX_GEMM_CFG=$cfg; fi
  python tools/gemm_bench.py --batches 2 --iters 60 --only geglu,out64,qk64,proj,ff2 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02k/stages.txt
