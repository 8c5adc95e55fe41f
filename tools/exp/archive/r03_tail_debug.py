import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from minddiffusion_amd import ops
from minddiffusion_amd.pipeline import DiffusionPipeline
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["sd2_512"]
rs = np.random.RandomState
ops.set_option("unet_st_tail", 32)
model = bench.build_model(dev, cfg["unet"])
net = model.model.diffusion_model if hasattr(model, "model") else model.unet
net.use_graph = False
pipe = DiffusionPipeline(model, sampler=cfg["sampler"], device=dev)
c = torch.from_numpy(rs(1).randn(1, 77, 1024).astype(np.float32)).to(dev, torch.float16)
uc = torch.from_numpy(rs(2).randn(1, 77, 1024).astype(np.float32)).to(dev, torch.float16)
x_T = torch.from_numpy(rs(42).randn(1, 4, 64, 64).astype(np.float32)).to(dev)
orig = net.forward_nhwc
calls = [0]
def checked(x, t, context, **kw):
    y = orig(x, t, context, **kw)
    torch.cuda.synchronize()
    calls[0] += 1
    if not torch.isfinite(y).all():
        print("non-finite UNet output at call", calls[0], "x absmax", float(x.abs().max()), "x finite", bool(torch.isfinite(x).all()), flush=True)
        P = net._plans[(x.shape[0], x.shape[2], x.shape[3])]
        for i, (op, m) in enumerate(zip(P.main[P.temb_ops:], P.meta[P.temb_ops:])):
            op()
        torch.cuda.synchronize()
        for k, td in enumerate(P.tails):
            o, tok, xin, out, kc, vtc = td._bufs
            st = {n: (bool(torch.isfinite(v).all()), float(v.float().abs().max())) for n, v in
                  (("attn_o", o), ("tok", tok), ("x_in", xin), ("out", out), ("kc", kc), ("vtc", vtc))}
            print("tail", k, st, flush=True)
            if st["attn_o"][0] and st["tok"][0] and st["x_in"][0] and not st["out"][0]:
                dbg = torch.zeros_like(out)
                for stage in (1, 2, 3, 4, 5, 6, 7):
                    td.debug_out, td.debug_stage = dbg.data_ptr(), stage
                    ops.st_tail_run(td)
                    torch.cuda.synchronize()
                    print("   stage", stage, "finite", bool(torch.isfinite(dbg).all()), "absmax", float(dbg.float().nan_to_num(0, 0, 0).abs().max()), flush=True)
                td.debug_out = 0
                break
        sys.exit(1)
    return y
net.forward_nhwc = checked
out = pipe(c=c, uc=uc, x_T=x_T, H=512, W=512, steps=50, scale=9.0, eta=0.0, decode=False, batch_size=1)
print("done finite", bool(torch.isfinite(out).all()), calls[0])
