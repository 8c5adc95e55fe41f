#!/bin/bash
# round-6 baseline on this round's box: default bench, eager op profile, gemm phase trace (the round-5 library as checked out)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06a
bash tools/gpu_pass.sh r06a bench opprof
timeout 200 python tools/gemm_trace.py > gpurun_out/r06a/gemm_trace.txt 2>&1; tail -30 gpurun_out/r06a/gemm_trace.txt
timeout 200 python tools/glide_op_profile.py > gpurun_out/r06a/glide_op_profile.txt 2>&1; head -20 gpurun_out/r06a/glide_op_profile.txt
