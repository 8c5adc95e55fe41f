set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
rm -f gpurun_out/parity_log.jsonl
timeout 600 python -m pytest tests/test_trajectories_gpu.py -m gpu -q 2>&1 | tail -4
cp gpurun_out/parity_log.jsonl gpurun_out/r04/parity_traj.jsonl
bash tools/gpu_pass.sh r04 bench prof profcfg configs opprof
