set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04i; mkdir -p $OUT
MDX_GEMM_DENSE8Q=0 timeout 200 python tools/op_profile.py --model wukong --batch 16 --top 250 > $OUT/op_q0.txt 2>&1
MDX_GEMM_DENSE8Q=1 MDX_GEMM_DENSE8Q_VAR=32 timeout 200 python tools/op_profile.py --model wukong --batch 16 --top 250 > $OUT/op_q1.txt 2>&1
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r'\s+#\s*(\d+)\s+(\w+)\s+([\d.]+) us\s+([\d.]+) TF/s\s+(.*)',l)
        if m: d[int(m.group(1))]=(float(m.group(3)),m.group(5).strip())
    return d
a=load('gpurun_out/r04i/op_q0.txt'); b=load('gpurun_out/r04i/op_q1.txt')
tot=0
for k in sorted(a):
    if k in b and abs(a[k][0]-b[k][0])>3:
        print(k, a[k][1], a[k][0], '->', b[k][0]); tot+=b[k][0]-a[k][0]
print('total delta us', tot)
PY
