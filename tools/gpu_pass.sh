#!/bin/bash
# One GPU-box pass (run from the repo root through gpurun): stages selected by name, outputs under gpurun_out/<tag>/.
#   tools/gpu_pass.sh <tag> tests bench configs prof profcfg pmc pmc_mfma opprof
set -u
export TMPDIR=/tmp
TAG=${1:-r02}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for stage in "$@"; do
  case $stage in
    tests)
      rm -f gpurun_out/parity_log.jsonl
      timeout 1500 python -m pytest tests -m gpu -q -rf --durations=15 > $OUT/pytest_gpu.log 2>&1
      tail -25 $OUT/pytest_gpu.log
      cp gpurun_out/parity_log.jsonl $OUT/parity.jsonl 2>/dev/null
      timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt ;;
    bench)
      timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err ;;
    configs)
      for cfg in wukong_512_plms sd2_768 glide_256 sd2_512_e2e; do
        timeout 400 python bench.py --config $cfg --no-cpu-baseline --steps 2 > $OUT/bench_$cfg.json 2>> $OUT/bench.err
        python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$cfg.json"))
    print("$cfg", d["value"], d["unit"], "gemm", d["roofline"]["achieved"], "whole", d["roofline"]["whole_path"]["achieved"])
except Exception as e:
    print("$cfg FAILED", e)
PY
      done ;;
    prof)
      timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o $TAG -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_prof.log 2>&1
      DB=$(find gpurun_out/prof_bench -name "*_results.db" | head -1)
      [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/bench_kernel_stats.md
      rm -rf gpurun_out/prof_bench; head -30 $OUT/bench_kernel_stats.md ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        for attempt in 1 2 3; do     # rocprofv3 --pmc occasionally dies with SIGSEGV inside the profiled process (ROCm 7.2): retry
          rm -rf gpurun_out/pmc/$c
          timeout 400 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc/$c -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-other-configs > $OUT/pmc_$c.log 2>&1
          [ -n "$(find gpurun_out/pmc/$c -name '*_results.db' 2>/dev/null | head -1)" ] && break
          echo "pmc $c attempt $attempt failed"
        done
        tail -2 $OUT/pmc_$c.log | cut -c1-200
      done
      python tools/pmc_traffic.py gpurun_out/pmc "python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph (the timed configuration launched eagerly: rocprofv3 --pmc segfaults on hipGraph replay; same kernels, same launch mix)" ${ATTN_PER_EVAL:-27} > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err; rm -rf gpurun_out/pmc; cat $OUT/pmc_traffic.json | head -60; cat $OUT/pmc_traffic.err | tail -3 ;;
    pmc_mfma)
      for attempt in 1 2 3; do
        rm -rf gpurun_out/pmc/MFMA
        timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc/MFMA -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-other-configs > $OUT/pmc_mfma.log 2>&1
        [ -n "$(find gpurun_out/pmc/MFMA -name '*_results.db' 2>/dev/null | head -1)" ] && break
        echo "pmc mfma attempt $attempt failed"
      done
      python tools/pmc_mfma.py $(find gpurun_out/pmc/MFMA -name '*_results.db' | head -1) "python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-other-configs (the timed configuration launched eagerly: rocprofv3 --pmc segfaults on hipGraph replay)" ${ATTN_PER_EVAL:-27} > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err; rm -rf gpurun_out/pmc/MFMA; cat $OUT/pmc_mfma.json | head -70; tail -3 $OUT/pmc_mfma.err ;;
    pmc_mfma_cfg)     # matrix-pipe occupancy of the other LDM configs (one evaluation each, eager)
      for cfg in wukong_512_plms sd2_768; do
        for attempt in 1 2 3; do
          rm -rf gpurun_out/pmc/MFMA_$cfg
          timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc/MFMA_$cfg -o pmc -- python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $OUT/pmc_mfma_$cfg.log 2>&1
          [ -n "$(find gpurun_out/pmc/MFMA_$cfg -name '*_results.db' 2>/dev/null | head -1)" ] && break
          echo "pmc mfma $cfg attempt $attempt failed"
        done
        python tools/pmc_mfma.py $(find gpurun_out/pmc/MFMA_$cfg -name '*_results.db' | head -1) "python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-graph (launched eagerly: rocprofv3 --pmc segfaults on hipGraph replay)" 27 > $OUT/pmc_mfma_$cfg.json 2> $OUT/pmc_mfma_$cfg.err; rm -rf gpurun_out/pmc/MFMA_$cfg; python -c "
import json; d=json.load(open('$OUT/pmc_mfma_$cfg.json')); print('$cfg', {k: (v['launches_per_eval'], v['mfma_busy']) for k, v in d['families'].items()})"
      done ;;
    profcfg)     # kernel traces of the other BASELINE configs (one unit each)
      for cfg in wukong_512_plms sd2_768 glide_256; do
        timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg -o $TAG -- python bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${cfg}_prof.log 2>&1
        DB=$(find gpurun_out/prof_cfg -name "*_results.db" | head -1)
        [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/${cfg}_kernel_stats.md
        rm -rf gpurun_out/prof_cfg; head -8 $OUT/${cfg}_kernel_stats.md | cut -c1-160
      done ;;
    opprof)
      timeout 200 python tools/op_profile.py --batch 2 --top 400 > $OUT/op_profile_b2.txt 2>&1; head -8 $OUT/op_profile_b2.txt ;;
    *) echo "unknown stage $stage" ;;
  esac
done
