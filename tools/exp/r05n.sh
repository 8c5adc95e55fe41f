#!/bin/bash
# round 5, GPU pass N: software-pipelined K loop on the generic kernel's small tiles (pre = the build before)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05n
mkdir -p $OUT
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
files=""
for lib in pre new; do
  path=$PWD/minddiffusion_amd/libmdx_$lib.so; [ $lib = new ] && path=$PWD/minddiffusion_amd/libmdx.so
  MDX_LIBRARY=$path timeout 200 python tools/op_profile.py --batch 2 --latent 64 --passes 7 --top 0 --out $OUT/ops_$lib.json 2>&1 | grep -v amdgpu.ids | head -1
  files="$files $OUT/ops_$lib.json"
done
python tools/exp/r05_opdiff.py $files | tee $OUT/opdiff.txt
for lib in pre new pre new; do
  path=$PWD/minddiffusion_amd/libmdx_$lib.so; [ $lib = new ] && path=$PWD/minddiffusion_amd/libmdx.so
  MDX_LIBRARY=$path timeout 200 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --rounds 5 --iters 20 --arms "$lib:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
for m in "wukong 16 64" "sd2 8 96"; do set -- $m
  for lib in pre new; do
    path=$PWD/minddiffusion_amd/libmdx_$lib.so; [ $lib = new ] && path=$PWD/minddiffusion_amd/libmdx.so
    MDX_LIBRARY=$path timeout 200 python tools/eval_ab.py --model $1 --batch $2 --latent $3 --rounds 3 --iters 5 --arms "$lib:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
  done
done
