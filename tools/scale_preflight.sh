#!/bin/bash
# Pre-flight of the multi-GPU scaling run on a node with >= 2 MI355X (the driver runs the real one at round end): bench.py at
# N = 1, 2, 4, 8 back to back, headline config, RCCL ("nccl") over xGMI; checks the self-description of every N > 1 line (backend,
# world size, ONE broadcast per trajectory) and weak-scaling efficiency value(N) >= 0.97 x N x value(1); then BASELINE configs[3] at
# 8 GPUs and configs[4] at 2 GPUs -- the GPU counts BASELINE.json quotes them on -- the same way.
#   tools/scale_preflight.sh [max_gpus]
set -u
MAXN=${1:-8}
OUT=gpurun_out/scale_preflight
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus 1 --steps 5 --warmup 1 --no-other-configs --no-cpu-baseline > $OUT/n1.json || exit 1
for N in 2 4 8; do
  [ $N -le $MAXN ] || break
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps 5 --warmup 1 --no-other-configs > $OUT/n$N.json || exit 1
done
# the configs BASELINE.json quotes on MORE than one GPU, at the GPU counts it quotes them on: configs[3] (SDv2 768x768, batch 32 over
# 8 GPUs = 4 per GPU) and configs[4] (Taichu-GLIDE, batch 16 over 2 GPUs = 8 per GPU); each against its own N = 1 line
for spec in "sd2_768 8" "glide_256 2"; do
  set -- $spec; CFG=$1; N=$2
  [ $N -le $MAXN ] || continue
  python bench.py --config $CFG --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${CFG}_n1.json || exit 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      bench.py --config $CFG --gpus $N --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${CFG}_n$N.json || exit 1
done
python - "$OUT" "$MAXN" <<'PY'
import json, sys
out, maxn = sys.argv[1], int(sys.argv[2])
base = json.load(open(f"{out}/n1.json"))["value"]
ok = True
for n in (2, 4, 8):
    if n > maxn:
        break
    d = json.load(open(f"{out}/n{n}.json"))
    c = d["config"]
    eff = d["value"] / (n * base)
    good = (d["n_gpus"] == n and c["dist_backend"] == "nccl" and c["rccl_world_size"] == n and c["broadcasts_per_step"] == 1
            and eff >= 0.97)
    ok &= good
    print(f"N={n}: {d['value']:.3f} {d['unit']}  efficiency {eff:.3f}  backend {c['dist_backend']} world {c['rccl_world_size']} "
          f"broadcast {c['broadcast_bytes']} B in {c['broadcast_ms']} ms  per-rank {c['per_rank_units_per_s']}  {'ok' if good else 'FAIL'}")
import os
for cfg, n in (("sd2_768", 8), ("glide_256", 2)):
    if n > maxn or not os.path.exists(f"{out}/{cfg}_n{n}.json"):
        continue
    b1 = json.load(open(f"{out}/{cfg}_n1.json"))["value"]
    d = json.load(open(f"{out}/{cfg}_n{n}.json"))
    c = d["config"]
    eff = d["value"] / (n * b1)
    good = d["n_gpus"] == n and c["dist_backend"] == "nccl" and c["rccl_world_size"] == n and c["broadcasts_per_step"] == 1 and eff >= 0.97
    ok &= good
    print(f"{cfg} N={n}: {d['value']:.3f} {d['unit']} (N=1 {b1:.3f})  efficiency {eff:.3f}  backend {c['dist_backend']} "
          f"broadcast {c['broadcast_bytes']} B in {c['broadcast_ms']} ms  {'ok' if good else 'FAIL'}")
sys.exit(0 if ok else 1)
PY
