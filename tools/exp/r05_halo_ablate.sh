#!/bin/bash
# round 5: timing ablations of the HALO conv's tap loop (what does a tap wait for when a block has its CU to itself?).
# Builds: minddiffusion_amd/libmdx_abl{1..4}.so = the diagnostics (trace) build of gemm.hip with -DMDX_HALO_ABLATE=n
# (1 no MFMAs, 2 no fragment reads, 3 no DMA issue inside the loop, 4 no barrier; WRONG RESULTS) + libmdx_trace.so = the product loop.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_ablate
mkdir -p $OUT
for lib in trace abl1 abl2 abl3 abl4; do
  for mode in "" "--warm"; do
    echo "== $lib ${mode:-cold}" | tee -a $OUT/halo_ablate.txt
    MDX_LIBRARY=$PWD/minddiffusion_amd/libmdx_$lib.so timeout 200 python tools/gemm_trace.py --only conv64_320_320,conv32_640_640,conv16_1280_1280 $mode 2>&1 | grep -E 'blocks; kernel|per-block' | tee -a $OUT/halo_ablate.txt
  done
done
