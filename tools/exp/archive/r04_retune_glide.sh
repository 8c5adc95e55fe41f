set -u
mkdir -p gpurun_out/tune
cp minddiffusion_amd/libmdx.so minddiffusion_amd/libmdx_old.so
T=gpurun_out/tune/gemm_tuned_glide.inc
cp minddiffusion_amd/csrc/gemm_tuned.inc $T
timeout 600 python tools/tune_gemm.py --model glide --merge --gain 0.03 --reps 5 --out $T --log gpurun_out/tune/insitu_glide.log 2>&1 | grep "KEEP\|entries"
cp $T minddiffusion_amd/csrc/gemm_tuned.inc
make -C minddiffusion_amd/csrc -j16 2>&1 | grep -E "error|Error"
OLD=$PWD/minddiffusion_amd/libmdx_old.so
for v in old new old new; do
  if [ $v = old ]; then L=$OLD; else L=$PWD/minddiffusion_amd/libmdx.so; fi
  MDX_LIBRARY=$L timeout 300 python bench.py --config glide_256 --no-cpu-baseline --steps 2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('glide_256 $v', r['value'])"
done
