#!/bin/bash
# round 6, pass P: attn_pipe_kernel with the packed fma through inline asm vs the builtin (split by the compiler), against attn_kernel; many interleaved rounds
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06p
mkdir -p $OUT
make -C minddiffusion_amd/csrc -j32 variant NAME=nopk EXTRA=-DMDX_ATTN_PIPE_PKASM=0 > $OUT/build.log 2>&1
A=$PWD/minddiffusion_amd/libmdx.so; Bv=$PWD/minddiffusion_amd/libmdx_nopk.so
for lib in asm builtin asm builtin; do
  path=$A; [ $lib = builtin ] && path=$Bv
  echo "== pk_fma: $lib" | tee -a $OUT/attn_bench.txt
  MDX_LIBRARY=$path timeout 300 python tools/attn_bench.py --shapes "8,5,9216,64;2,5,4096,64;16,8,4096,40;16,10,1024,64" --forms o3,p,o3s,ps --iters 10 --rounds 9 2>&1 | grep -v amdgpu | tee -a $OUT/attn_bench.txt
done
