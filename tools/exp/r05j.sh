#!/bin/bash
# round 5, GPU pass J: kernel-argument warm-up in every parameter-struct kernel -- parity gates, then per-config bench of the round-4
# library (old), the build before the warm-up (prekt) and the new one, alternating on one box; HIP_FORCE_DEV_KERNARG probe
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05j
mkdir -p $OUT
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_conv8p_gpu.py tests/test_stchain_gpu.py tests/test_glide_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for c in sd2_512 wukong_512_plms sd2_768 glide_256; do
  for v in old prekt new old prekt new; do
    L=$PWD/minddiffusion_amd/libmdx_$v.so; [ $v = old ] && L=$PWD/minddiffusion_amd/libmdx_base.so; [ $v = new ] && L=$PWD/minddiffusion_amd/libmdx.so
    MDX_LIBRARY=$L timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', r['value'], r.get('per_unet_step_ms'))" | tee -a $OUT/bench_ab.txt
  done
done
for e in 0 1; do
  HIP_FORCE_DEV_KERNARG=$e timeout 300 python bench.py --config sd2_512 --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sd2_512 new HIP_FORCE_DEV_KERNARG=$e', r['value'], r.get('per_unet_step_ms'))" | tee -a $OUT/bench_ab.txt
  HIP_FORCE_DEV_KERNARG=$e MDX_LIBRARY=$PWD/minddiffusion_amd/libmdx_base.so timeout 300 python bench.py --config sd2_512 --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sd2_512 old HIP_FORCE_DEV_KERNARG=$e', r['value'], r.get('per_unet_step_ms'))" | tee -a $OUT/bench_ab.txt
done
