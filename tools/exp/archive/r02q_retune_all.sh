#!/bin/bash
# retune all benchmarked configs with the LDS ring depth as a tuned dimension (and the in-kernel split-K reduce), rebuild, bench
set -u
export TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/${TAG:-r02q}; mkdir -p $OUT
cp minddiffusion_amd/csrc/gemm_tuned.inc $OUT/gemm_tuned.inc
t0=$(date +%s)
timeout 600 python tools/tune_gemm.py --reps ${REPS:-7} --model sd2 --batch 2 --latent 64 --merge --out $OUT/gemm_tuned.inc --log $OUT/sd2_b2.log 2>&1 | tail -1
echo "sd2 b2 done $(( $(date +%s) - t0 )) s"
timeout 600 python tools/tune_gemm.py --reps ${REPS:-7} --model wukong --batch 16 --latent 64 --merge --out $OUT/gemm_tuned.inc --log $OUT/wukong_b16.log 2>&1 | tail -1
echo "wukong b16 done $(( $(date +%s) - t0 )) s"
timeout 600 python tools/tune_gemm.py --reps ${REPS:-7} --model sd2 --batch 8 --latent 96 --merge --out $OUT/gemm_tuned.inc --log $OUT/sd2_b8_l96.log 2>&1 | tail -1
echo "sd2 b8 l96 done $(( $(date +%s) - t0 )) s"
timeout 700 python tools/tune_gemm.py --reps ${REPS:-7} --model glide --merge --out $OUT/gemm_tuned.inc --log $OUT/glide.log 2>&1 | tail -1
echo "glide done $(( $(date +%s) - t0 )) s"
wc -l $OUT/gemm_tuned.inc
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('before', r['value'], r['per_unet_step_ms'])"
cp $OUT/gemm_tuned.inc minddiffusion_amd/csrc/gemm_tuned.inc
(cd minddiffusion_amd/csrc && make 2>&1 | grep -E "error|Error")
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('after', r['value'], r['per_unet_step_ms'], r['roofline']['launches_per_unit_of_profile'])"
for cfg in wukong_512_plms sd2_768 glide_256; do
  timeout 400 python bench.py --config $cfg --no-cpu-baseline --steps 2 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', r['value'], r['unit'])"
done
