#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06h
timeout 1200 python -m pytest tests/test_glide_gpu.py -m gpu -x -q > gpurun_out/r06h/pytest_glide.log 2>&1; tail -6 gpurun_out/r06h/pytest_glide.log
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_trajectories_gpu.py -m gpu -x -q -k "glide" > gpurun_out/r06h/pytest_glide_cfg.log 2>&1; tail -6 gpurun_out/r06h/pytest_glide_cfg.log
for arm in 0 1 0 1; do
  MDX_GLIDE_QKV_MERGE=$arm timeout 400 python bench.py --config glide_256 --no-cpu-baseline --steps 3 > gpurun_out/r06h/bench_glide_merge$arm.json 2>> gpurun_out/r06h/bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r06h/bench_glide_merge$arm.json"))
print("merge=$arm", d["value"], d["unit"], {k: (v["ms"], v["launches"]) for k, v in d["roofline"]["families"].items()})
PY
done
