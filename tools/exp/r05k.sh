#!/bin/bash
# round 5, GPU pass K: GroupNorm-from-split-K slabs summed two pixels at a time, all loads of a batch in flight (pre = the build before)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05k
mkdir -p $OUT
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -m gpu -q -x -k "groupnorm or splitk or tiny_unet_forward or sd2_full_size_single_step" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
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
