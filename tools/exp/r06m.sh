#!/bin/bash
# round 6, pass M: batched store loops (EMODE 2) of the HALO patch tiles vs the generic per-pass loops -- kernel parity tests, then
# whole-evaluation A/B (alternating processes, one box) against a -DMDX_LEAN_EPI=0 build made here, phase traces.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06m
mkdir -p $OUT
make -C minddiffusion_amd/csrc -j32 variant NAME=epi0 EXTRA=-DMDX_HALO_LEAN_EPI=0 > $OUT/build_epi0.log 2>&1 &
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_conv8p_gpu.py tests/test_unet_gpu.py tests/test_glide_gpu.py -m gpu -x -q > $OUT/pytest_kernels.log 2>&1; tail -3 $OUT/pytest_kernels.log
wait
ls -la minddiffusion_amd/*.so
OLD=$PWD/minddiffusion_amd/libmdx_epi0.so; NEW=$PWD/minddiffusion_amd/libmdx.so
for cfg in "sd2 2 64" "sd2 8 96" "wukong 16 64"; do
  set -- $cfg
  for lib in old new old new; do
    path=$OLD; [ $lib = new ] && path=$NEW
    MDX_LIBRARY=$path timeout 300 python tools/eval_ab.py --model $1 --batch $2 --latent $3 --rounds 5 --iters 20 --arms "$lib:" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
  done
done
python tools/gemm_trace.py --batch 18 --only conv32,conv64 2>&1 | grep -v "amdgpu\|warning\|\^\|^ *[0-9]* |" > $OUT/gemm_trace_b18.txt
python tools/gemm_trace.py --batch 2 --only conv16,conv32,conv64 2>&1 | grep -v "amdgpu\|warning\|\^\|^ *[0-9]* |" > $OUT/gemm_trace_b2.txt
grep "phase durations\|^proj\|^ff2\|^geglu" $OUT/gemm_trace_b18.txt $OUT/gemm_trace_b2.txt
