#!/bin/bash
# round 6: Taichu-GLIDE whole-loop tables -- tests, then same-box A/B of the glide_256 bench config (tables off / on)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_glide_gpu.py tests/test_unet_gpu.py -m gpu -x -q -k "glide or inpaint or loop_tables or kv_select" > gpurun_out/r06b/pytest.log 2>&1; tail -15 gpurun_out/r06b/pytest.log
for arm in 0 1 0 1; do
  MDX_GLIDE_LOOP_TABLES=$arm timeout 400 python bench.py --config glide_256 --no-cpu-baseline --steps 3 > gpurun_out/r06b/bench_glide_tables$arm.json 2>> gpurun_out/r06b/bench.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r06b/bench_glide_tables$arm.json"))
print("tables=$arm", d["value"], d["unit"], {k: (v["ms"], v["launches"]) for k, v in d["roofline"]["families"].items()}, d["config"].get("loop_prefix_ms"))
PY
done
tail -5 gpurun_out/r06b/bench.err
