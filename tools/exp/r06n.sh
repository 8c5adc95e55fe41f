#!/bin/bash
# round 6, pass N: software-pipelined attention (attn_pipe_kernel) -- parity, micro-benchmark, whole-evaluation A/B (option arms, one process)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06n
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > $OUT/pytest_attn.log 2>&1; tail -4 $OUT/pytest_attn.log
timeout 600 python tools/attn_bench.py --shapes "8,5,9216,64;2,5,4096,64;16,8,4096,40;16,10,1024,64;16,8,1024,80;16,4,4224,64;8,5,2304,64" --forms o3,o3s,p,ps --iters 10 2>&1 | grep -v amdgpu | tee $OUT/attn_bench.txt
for cfg in "sd2 2 64" "sd2 8 96" "wukong 16 64"; do
  set -- $cfg
  timeout 300 python tools/eval_ab.py --model $1 --batch $2 --latent $3 --rounds 5 --iters 20 --arms "base:attn_pipe=0" "pipe:attn_pipe=1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
