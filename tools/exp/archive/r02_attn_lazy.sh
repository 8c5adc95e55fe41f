#!/bin/bash
# attention: reference maximum moved only when a query outgrows it by more than 2^8 (lazy rescale)
export PYTHONPATH=.
mkdir -p gpurun_out/r02lazy
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or attn" 2>&1 | grep -v amdgpu.ids | tail -3
python tools/attn_bench.py --iters 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02lazy/attn_bench.txt
