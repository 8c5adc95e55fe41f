#!/bin/bash
# round 5, GPU pass O: GLIDE -- SiLU of the time embedding applied once (on the way out of its producer)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05o
mkdir -p $OUT
timeout 500 python -m pytest tests/test_glide_gpu.py tests/test_configs_gpu.py tests/test_trajectories_gpu.py -m gpu -q -x -k "glide or config4" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for i in 1 2; do
timeout 300 python bench.py --config glide_256 --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('glide_256 new', r['value'], r['roofline']['families'].get('small'))" | tee -a $OUT/bench.txt
done
