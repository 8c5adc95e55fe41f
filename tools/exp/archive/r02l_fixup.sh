#!/bin/bash
# in-kernel split-K reduce (last block of a tile sums the partials) vs slabs + reduce kernel
export PYTHONPATH=.
mkdir -p gpurun_out/r02l
echo "== kernel tests (fixup on)"
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -5
echo "== kernel tests (fixup off)"
MDX_GEMM_SPLITK_FIXUP=0 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "split or gemm or conv" 2>&1 | grep -v amdgpu.ids | tail -3
for f in 1 0; do
  echo "-- MDX_GEMM_SPLITK_FIXUP=$f"
  MDX_GEMM_SPLITK_FIXUP=$f python tools/gemm_bench.py --batches 2 --iters 60 --only conv32_1280,conv16,conv8,ff2,proj8,conv32_640 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02l/micro.txt
for f in 1 0; do
  MDX_GEMM_SPLITK_FIXUP=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02l/bench_fix$f.json 2> gpurun_out/r02l/bench_fix$f.err
done
python - <<'PY'
import json
for f in (1, 0):
    try:
        r = json.loads(open(f"gpurun_out/r02l/bench_fix{f}.json").read().strip().splitlines()[-1])
        print("fixup", f, r["value"], r["ms_per_step"], r.get("roofline", {}).get("launches"))
    except Exception as e:
        print("fixup", f, "failed", e); print(open(f"gpurun_out/r02l/bench_fix{f}.err").read()[-1500:])
PY
