#!/bin/bash
# round-2 experiment pass c: colstats GroupNorm, fused reduce+GN, HALO8 tuning, rolled generic kernel at batch 2
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02c; mkdir -p $OUT
echo "== kernel tests"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q 2>&1 | tail -5
echo "== unet tests (tiny + full size single steps)"
timeout 900 python -m pytest tests/test_unet_gpu.py -q -k "not ddim_cfg_trajectory" 2>&1 | tail -5
echo "== bench A/B"
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_$name.json 2>$OUT/bench_$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/bench_$name.json')); f=d['roofline']['families']
    print('$name', d['value'], d['per_unet_step_ms'], 'gemm', f['gemm']['ms'], f['gemm']['launches'], 'gn', f['groupnorm']['ms'], f['groupnorm']['launches'], 'attn', f['attention']['ms'])
except Exception as e:
    print('$name FAILED', e); print(open('$OUT/bench_$name.err').read()[-600:])
"; }
run default X=1
run no_colstats MDX_UNET_GN_COLSTATS=0
run no_fuse MDX_UNET_GN_SPLITK_FUSE=0
run neither MDX_UNET_GN_COLSTATS=0 MDX_UNET_GN_SPLITK_FUSE=0
run halo8_off MDX_GEMM_HALO8=0
run roll MDX_GEMM_ROLL=1
echo "== tune 8x8 convs (HALO8 vs generic)"
timeout 300 python tools/tune_gemm.py --model sd2 --batch 2 --latent 64 --only-m 128 --only-ks 3 --merge --out $OUT/tuned_halo8.inc --log $OUT/tune_halo8.log 2>&1 | tail -8
cat $OUT/tuned_halo8.inc | tail -5
