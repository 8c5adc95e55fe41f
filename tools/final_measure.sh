#!/bin/bash
# Round-end measurement pass on the GPU box (run from the repo root through gpurun):
# PMC traffic passes -> full GPU test suite -> default bench line -> rocprofv3 kernel trace of the bench command
# -> the other BASELINE configs.  Everything judged is written under gpurun_out/final/ (copied to profiles/ afterwards).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/final
mkdir -p $OUT
TAG=${1:-r01_p}

for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc/$c -o pmc -- python tools/op_profile.py --batch 2 --passes 1 > $OUT/pmc_$c.log 2>&1
done
python tools/pmc_traffic.py gpurun_out/pmc > $OUT/${TAG}_pmc_traffic.json 2>$OUT/pmc_traffic.err && cp $OUT/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
rm -rf gpurun_out/pmc

timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > $OUT/smoke.txt 2>&1
timeout 300 python bench.py > $OUT/${TAG}_bench.json 2>$OUT/bench.err
timeout 100 python tools/op_profile.py --batch 2 > $OUT/${TAG}_op_profile_b2.txt 2>&1

timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o ${TAG} -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_prof.log 2>&1
DB=$(find gpurun_out/prof_bench -name "*_results.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/${TAG}_bench_kernel_stats.md
rm -rf gpurun_out/prof_bench

for cfg in wukong_512_plms sd2_768 glide_256 sd2_512_e2e sd2_512_dpm_solver; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline --steps 2 > $OUT/${TAG}_bench_$cfg.json 2>>$OUT/bench.err
done
tail -3 $OUT/pytest_gpu.txt; cat $OUT/${TAG}_bench.json
