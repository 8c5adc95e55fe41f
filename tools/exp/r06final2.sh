#!/bin/bash
# round 6, last pass on the final tree: suite + smoke, default bench line (reads the re-collected PMC files)
set -u
export TMPDIR=/tmp
bash tools/gpu_pass.sh r06k tests bench
