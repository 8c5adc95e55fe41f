#!/bin/bash
# how much faster are the weight-heavy UNet-batch-2 launches when their weights come from the Infinity Cache instead of HBM?
export PYTHONPATH=.
mkdir -p gpurun_out/r02o
for mode in cold hot; do
  echo "-- $mode"
  if [ $mode = hot ]; then H=--hot; else H=; fi
  python tools/gemm_bench.py --batches 2 --iters 60 $H --only conv32_1280,conv16,conv8,ff2_16,ff2_32,proj8,proj16,geglu16,geglu32,qk16 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02o/hot_cold.txt
