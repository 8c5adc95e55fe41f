#!/bin/bash
# round 6, pass Y: fused cross-attention on 64-row tiles (n % 64 == 0) -- parity; GroupNorm geometry options on Taichu-GLIDE (alternating processes)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06y
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "cross_attention_epilogue" > $OUT/pytest_k.log 2>&1; tail -3 $OUT/pytest_k.log
timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "config3 or ragged or sweep" > $OUT/pytest_c.log 2>&1; tail -3 $OUT/pytest_c.log
timeout 300 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --rounds 5 --iters 10 --arms "base:unet_xattn_fuse=0" "fuse:unet_xattn_fuse=1" "base2:unet_xattn_fuse=0" "fuse2:unet_xattn_fuse=1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
run() { name=$1; shift
  env "$@" timeout 400 python bench.py --config glide_256 --no-cpu-baseline --steps 3 > $OUT/bench_glide_$name.json 2>> $OUT/bench.err
  python -c "
import json; d=json.load(open('$OUT/bench_glide_$name.json')); print('glide $name', d['value'], {k:(round(v['ms'],1),v['launches']) for k,v in d['roofline']['families'].items()})"
}
run base X=0
run gnfused0 MDX_GN_FUSED=0
run boost10 MDX_GN_BOOST_MB=10
run chunks8 MDX_GN_COL_CHUNKS=8
run minblk1024 MDX_GN_MIN_BLOCKS=1024
run base2 X=0
