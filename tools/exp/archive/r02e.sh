#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_glide_gpu.py tests/test_vae_gpu.py tests/test_text_encoder_gpu.py -q 2>&1 | tail -4
MDX_ATTN_NW=4 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "attention" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_configs_gpu.py -q -k "config2 or config3 or config4" 2>&1 | tail -4
run() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$name.json 2>$OUT/bench_$name.err; python -c "
import json
try:
    d=json.load(open('$OUT/bench_$name.json')); f=d['roofline']['families']
    print('$name', d['value'], d.get('per_unet_step_ms'), {k:(v['ms'],v['launches']) for k,v in f.items()})
except Exception as e:
    print('$name FAILED', e); print(open('$OUT/bench_$name.err').read()[-600:])
"; }
run sd2_nw_auto sd2_512 X=1
run sd2_nw4 sd2_512 MDX_ATTN_NW=4
run sd2_nw2 sd2_512 MDX_ATTN_NW=2
run glide_cs glide_256 X=1
run glide_nocs glide_256 MDX_UNET_GN_COLSTATS=0
run wukong_cs wukong_512_plms X=1
run wukong_nocs wukong_512_plms MDX_UNET_GN_COLSTATS=0
