#!/bin/bash
# round 6, pass Q: attn_pipe default on -- GPU suites that touch attention + whole-evaluation A/B (option arms, one process)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06q
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_glide_gpu.py tests/test_stchain_gpu.py tests/test_fp16_reference_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for cfg in "sd2 2 64" "sd2 8 96" "wukong 16 64"; do
  set -- $cfg
  timeout 300 python tools/eval_ab.py --model $1 --batch $2 --latent $3 --rounds 7 --iters 20 --arms "base:attn_pipe=0" "pipe:attn_pipe=1" "base2:attn_pipe=0" "pipe2:attn_pipe=1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
for arm in 0 1 0 1; do
  MDX_ATTN_PIPE=$arm timeout 400 python bench.py --config glide_256 --no-cpu-baseline --steps 3 > $OUT/bench_glide_pipe$arm.json 2>> $OUT/bench.err
  python -c "
import json; d=json.load(open('$OUT/bench_glide_pipe$arm.json')); print('glide attn_pipe=$arm', d['value'], {k:(round(v['ms'],1),v['launches']) for k,v in d['roofline']['families'].items()})"
done
