#!/bin/bash
# round 5, GPU pass E: deeper LDS rings under the lean dense K loop (library = new gemm loop + round-4 HALO loop)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05e
mkdir -p $OUT
export MDX_LIBRARY=$PWD/minddiffusion_amd/libmdx_h0.so
timeout 300 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --rounds 5 --iters 20 \
   --arms "h0:" "ring4:gemm_ring=4" "ring5:gemm_ring=5" "untuned:gemm_tuned=0" "untuned_ring4:gemm_tuned=0,gemm_ring=4" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_ring.txt
