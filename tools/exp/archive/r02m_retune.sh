#!/bin/bash
# retune the UNet-batch-2 tile table with the in-kernel split-K reduce (<= 4 splits) available, rebuild, bench
set -u
export TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02m; mkdir -p $OUT
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_before.json 2> $OUT/bench_before.err
cp minddiffusion_amd/csrc/gemm_tuned.inc $OUT/gemm_tuned.inc
t0=$(date +%s)
timeout 500 python tools/tune_gemm.py --model sd2 --batch 2 --latent 64 --merge --out $OUT/gemm_tuned.inc --log $OUT/sd2_b2.log 2>&1 | tail -2
echo "sd2 b2 done $(( $(date +%s) - t0 )) s"
cp $OUT/gemm_tuned.inc minddiffusion_amd/csrc/gemm_tuned.inc
(cd minddiffusion_amd/csrc && make 2>&1 | grep -E "error|Error" ; ls -la ../libmdx.so)
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_after.json 2> $OUT/bench_after.err
python - <<'PY'
import json
for f in ("before", "after"):
    try:
        r = json.loads(open(f"gpurun_out/r02m/bench_{f}.json").read().strip().splitlines()[-1])
        print(f, r["value"], r["ms_per_step"], r.get("roofline", {}))
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/r02m/bench_{f}.err").read()[-1500:])
PY
