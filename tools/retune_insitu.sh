set -u
mkdir -p gpurun_out/tune
cp minddiffusion_amd/libmdx.so minddiffusion_amd/libmdx_old.so
T=gpurun_out/tune/gemm_tuned_insitu.inc
cp minddiffusion_amd/csrc/gemm_tuned.inc $T
timeout 300 python tools/tune_gemm.py --model wukong --batch 16 --latent 64 --merge --gain 0.03 --reps 5 --out $T --log gpurun_out/tune/insitu_wukong_b16.log 2>&1 | grep "KEEP\|entries"
timeout 300 python tools/tune_gemm.py --model sd2 --batch 8 --latent 96 --merge --gain 0.03 --reps 5 --out $T --log gpurun_out/tune/insitu_sd2_b8_l96.log 2>&1 | grep "KEEP\|entries"
timeout 400 python tools/tune_gemm.py --model glide --merge --gain 0.03 --reps 5 --out $T --log gpurun_out/tune/insitu_glide.log 2>&1 | grep "KEEP\|entries"
timeout 300 python tools/tune_gemm.py --model sd2 --batch 2 --latent 64 --merge --gain 0.03 --reps 9 --out $T --log gpurun_out/tune/insitu_sd2_b2.log 2>&1 | grep "KEEP\|entries"
cp $T minddiffusion_amd/csrc/gemm_tuned.inc
make -C minddiffusion_amd/csrc -j16 2>&1 | grep -E "error|Error"
OLD=$PWD/minddiffusion_amd/libmdx_old.so
for c in wukong_512_plms sd2_768 sd2_512 glide_256; do
  for v in old new old new; do
    if [ $v = old ]; then L=$OLD; else L=$PWD/minddiffusion_amd/libmdx.so; fi
    MDX_LIBRARY=$L timeout 200 python bench.py --config $c --no-cpu-baseline --steps 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', r['value'], r.get('per_unet_step_ms'))"
  done
done
