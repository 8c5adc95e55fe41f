#!/bin/bash
# round 5, GPU pass D: per-op profile of one evaluation under several builds of the library (which launches moved?)
#   bash tools/exp/r05d.sh [batch latent model] -- libs: libmdx_base.so (round 4) + every libmdx_*.so variant present + libmdx.so
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05d
mkdir -p $OUT
B=${1:-2}; L=${2:-64}; M=${3:-sd2}
files=""
for lib in base $(ls minddiffusion_amd/ | sed -n 's/^libmdx_\(.*\)\.so$/\1/p' | grep -v -e '^base$' -e '^trace$') new; do
  path=$PWD/minddiffusion_amd/libmdx_$lib.so; [ $lib = new ] && path=$PWD/minddiffusion_amd/libmdx.so
  MDX_LIBRARY=$path timeout 200 python tools/op_profile.py --model $M --batch $B --latent $L --passes 7 --top 0 --out $OUT/ops_${M}_b${B}_$lib.json 2>&1 | grep -v amdgpu.ids | head -1
  files="$files $OUT/ops_${M}_b${B}_$lib.json"
done
python tools/exp/r05_opdiff.py $files | tee $OUT/opdiff_${M}_b${B}.txt
for lib in base new base new; do
  path=$PWD/minddiffusion_amd/libmdx_$lib.so; [ $lib = new ] && path=$PWD/minddiffusion_amd/libmdx.so
  MDX_LIBRARY=$path timeout 200 python tools/eval_ab.py --model $M --batch $B --latent $L --rounds 5 --iters 20 --arms "$lib:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_${M}_b${B}.txt
done
