#!/bin/bash
# rocprofv3 kernel traces of the other BASELINE configs' bench commands (graph replay), summarised like the headline's
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02prof; mkdir -p $OUT
for cfg in wukong_512_plms sd2_768 glide_256; do
  rm -rf gpurun_out/prof_cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg -o $cfg -- python bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$cfg.log 2>&1
  DB=$(find gpurun_out/prof_cfg -name "*_results.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/${cfg}_kernel_stats.md
  grep '"metric"' $OUT/$cfg.log | cut -c1-160
  head -8 $OUT/${cfg}_kernel_stats.md | cut -c1-200
done
rm -rf gpurun_out/prof_cfg
