import os, sys, math, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from minddiffusion_amd import ops
from oracle import ldm as O
DEV = "cuda:0"
B, H, W, Cin, Cout, splitk, rows = 2, 32, 32, 320, 320, 5, 128
rng = np.random.RandomState(B * H + Cin + Cout + rows)
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 0
x = (rng.standard_normal((B, Cin, H, W)) * (0.5 + rng.rand(Cin))[None, :, None, None] + (rng.standard_normal(Cin)[None, :, None, None] if MODE & 1 else 0)).astype(np.float16).astype(np.float32)
g = (1.0 + 0.2 * rng.standard_normal(Cin)).astype(np.float32) if MODE & 2 else np.ones(Cin, np.float32)
bt = (0.1 * rng.standard_normal(Cin)).astype(np.float32) if MODE & 4 else np.zeros(Cin, np.float32)
wt = (rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin)).astype(np.float16).astype(np.float32)
a = O.silu(O.group_norm(torch.tensor(x), torch.tensor(g), torch.tensor(bt), 1e-5))
xd = torch.tensor(np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(B, H * W, Cin))).to(DEV, torch.float16)
nrb = H * W // rows
blk = xd.float().reshape(B * nrb, rows, Cin)
cs = torch.stack([blk.sum(1), (blk * blk).sum(1)], 2).contiguous()
out = torch.empty((B, H * W, Cout), dtype=torch.float16, device=DEV)
wp = ops.pack_conv_weight(torch.tensor(wt).to(DEV))
for use_gn in (0, 1):
    src = xd if use_gn else torch.tensor(np.ascontiguousarray(a.numpy().transpose(0, 2, 3, 1).reshape(B, H * W, Cin))).to(DEV, torch.float16)
    kw = dict(gn_colstats=cs, gn_nrb=nrb, gn_gamma=torch.tensor(g, device=DEV), gn_beta=torch.tensor(bt, device=DEV)) if use_gn else {}
    d = ops.make_gemm_desc(src, wp, Cout, B, H, W, Cin, out, Cout, ksize=3, splitk=splitk, tile_n=64, **kw)
    ws = ops.new_gemm_workspace(64 << 20, DEV)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    print("query", ops.gemm_query(d))
    ops.gemm_run(d); torch.cuda.synchronize()
    slabs = ws[4096:4096 + splitk * B * H * W * Cout].reshape(splitk, B, H, W, Cout).cpu()
    for sp in range(splitk):
        ref = torch.nn.functional.conv2d(a[:, sp * 64:(sp + 1) * 64], torch.tensor(wt)[:, sp * 64:(sp + 1) * 64], padding=1).permute(0, 2, 3, 1)
        e = (slabs[sp] - ref).norm() / ref.norm()
        # error per sample and per patch row
        es = [(float((slabs[sp][b] - ref[b]).norm() / ref[b].norm())) for b in range(B)]
        print("gn" if use_gn else "plain", "split", sp, "rel", float(e), "per sample", es)
