#!/bin/bash
# Pre-flight of the multi-GPU scaling run on a node with >= 2 MI355X (the driver runs the real one at round end): bench.py at
# N = 1, 2, 4, 8 back to back, headline config, RCCL ("nccl") over xGMI; checks the self-description of every N > 1 line (backend,
# world size, ONE broadcast per trajectory) and weak-scaling efficiency value(N) >= 0.97 x N x value(1).
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
sys.exit(0 if ok else 1)
PY
