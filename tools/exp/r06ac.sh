#!/bin/bash
# round 6, pass AC: conv8p epilogue store loop in batches (MDX_C8_BATCHED_EPI) -- parity + alternating-process A/B against the per-pass loop
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06ac
mkdir -p $OUT
timeout 900 python -m pytest tests/test_conv8p_gpu.py -m gpu -x -q > $OUT/pytest_c8.log 2>&1; tail -3 $OUT/pytest_c8.log
make -C minddiffusion_amd/csrc -j16 variant NAME=c8old EXTRA=-DMDX_C8_BATCHED_EPI=0 2>&1 | grep -E "error|Error"
OLD=$PWD/minddiffusion_amd/libmdx_c8old.so
NEW=$PWD/minddiffusion_amd/libmdx.so
for v in old new old new; do
  if [ $v = old ]; then L=$OLD; else L=$NEW; fi
  echo "== $v"
  MDX_LIBRARY=$L timeout 200 python tools/eval_ab.py --guidance --model wukong --batch 16 --latent 64 --rounds 5 --iters 10 --arms "x:" 2>&1 | grep -v amdgpu.ids
  MDX_LIBRARY=$L timeout 200 python tools/eval_ab.py --guidance --model sd2 --batch 8 --latent 96 --rounds 5 --iters 10 --arms "x:" 2>&1 | grep -v amdgpu.ids
  MDX_LIBRARY=$L timeout 400 python bench.py --config glide_256 --no-cpu-baseline --steps 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('glide', r['value'], round(r['roofline']['families']['gemm']['ms'],1))"
done | tee $OUT/ab.txt
