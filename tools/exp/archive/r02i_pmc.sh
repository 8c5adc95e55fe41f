#!/bin/bash
# PMC diagnostic of the two representative batch-16 kernels: what does the L2 -> LDS operand path look like?
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02i; mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP_TCC_READ_REQ_LATENCY_sum|TCP_TCC_READ_REQ_sum|TCP_PENDING_STALL_CYCLES_sum|TCC_HIT_sum|TCC_MISS_sum|TCC_REQ_sum|TCC_EA0_RDREQ_sum|TA_TA_BUSY_sum|TA_BUSY_avr|TCP_GATE_EN1_sum|TCP_TA_TCP_STATE_READ_sum|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY|SQ_ACTIVE_INST_VMEM|SQ_ACTIVE_INST_LDS|SQ_INSTS_VMEM|SQ_INST_LEVEL_VMEM|SQ_VALU_MFMA_BUSY_CYCLES|SQ_INSTS_MFMA|GRBM_GUI_ACTIVE|TCP_TOTAL_CACHE_ACCESSES_sum|TCC_BUSY_sum|TCC_TAG_STALL_sum)\b" | sort -u > $OLDPWD/$OUT/available.txt
cd $OLDPWD
cat $OUT/available.txt | tr '\n' ' '; echo
KSETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum")
for case in "geglu:geglu64_320:gemm_kernel" "conv:conv64_320_320:conv3x3_halo"; do
  name=${case%%:*}; rest=${case#*:}; only=${rest%%:*}; kern=${rest#*:}
  for set in "${KSETS[@]}"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-50)
    timeout 200 rocprofv3 --pmc $set -d gpurun_out/pmci/$name/$tag -o pmc -- python tools/gemm_bench.py --batches 16 --only $only --iters 6 > $OUT/${name}_$tag.log 2>&1 || echo "pass $tag failed"
  done
  echo "== $name ($only, kernel $kern)"
  python tools/pmc_kernel.py $kern $(find gpurun_out/pmci/$name -name "*_results.db") | tee $OUT/${name}_pmc.txt
done
rm -rf gpurun_out/pmci
