#!/bin/bash
# round 6, pass O: attn_kernel before / after the attn_finish refactor (same box, alternating processes) + pipe
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06o
mkdir -p $OUT
OLD=$PWD/minddiffusion_amd/prev_attn_libmdx.so; NEW=$PWD/minddiffusion_amd/libmdx.so
for lib in old new old new; do
  path=$OLD; [ $lib = new ] && path=$NEW
  echo "== $lib" | tee -a $OUT/attn_bench.txt
  MDX_LIBRARY=$path timeout 300 python tools/attn_bench.py --shapes "8,5,9216,64;2,5,4096,64;16,8,4096,40;16,10,1024,64" --forms o3,o3s --iters 10 2>&1 | grep -v amdgpu | tee -a $OUT/attn_bench.txt
done
for cfg in "sd2 2 64" "sd2 8 96" "wukong 16 64"; do
  set -- $cfg
  for lib in old new old new; do
    path=$OLD; [ $lib = new ] && path=$NEW
    MDX_LIBRARY=$path timeout 300 python tools/eval_ab.py --model $1 --batch $2 --latent $3 --rounds 5 --iters 20 --arms "$lib:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
  done
done
