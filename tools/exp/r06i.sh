#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06i
run() { # name, env...
  name=$1; shift
  env "$@" timeout 400 python bench.py --config glide_256 --no-cpu-baseline --steps 3 > gpurun_out/r06i/bench_glide_$name.json 2>> gpurun_out/r06i/bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r06i/bench_glide_$name.json"))
print("$name", d["value"], d["unit"], {k: (round(v["ms"],1), v["launches"]) for k, v in d["roofline"]["families"].items()})
PY
}
timeout 600 python -m pytest tests/test_glide_gpu.py -m gpu -x -q -k "loop or unet" > gpurun_out/r06i/pytest_fuse.log 2>&1 &
wait
MDX_GLIDE_GN_QKV_FUSE=1 timeout 600 python -m pytest tests/test_glide_gpu.py -m gpu -x -q > gpurun_out/r06i/pytest_fuse.log 2>&1; tail -4 gpurun_out/r06i/pytest_fuse.log
run base X=0
run gnfuse MDX_GLIDE_GN_QKV_FUSE=1
run fold MDX_GN_COLSTATS_FOLD=1
run base2 X=0
run gnfuse2 MDX_GLIDE_GN_QKV_FUSE=1
run fold2 MDX_GN_COLSTATS_FOLD=1
tail -3 gpurun_out/r06i/bench.err
