import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from minddiffusion_amd import ops
from minddiffusion_amd.pipeline import DiffusionPipeline
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["sd2_512"]
rs = np.random.RandomState
ops.set_option("unet_st_tail", int(sys.argv[1]))
model = bench.build_model(dev, cfg["unet"])
net = model.model.diffusion_model if hasattr(model, "model") else model.unet
net.use_graph = bool(int(sys.argv[2]))
pipe = DiffusionPipeline(model, sampler=cfg["sampler"], device=dev)
c = torch.from_numpy(rs(1).randn(1, 77, 1024).astype(np.float32)).to(dev, torch.float16)
uc = torch.from_numpy(rs(2).randn(1, 77, 1024).astype(np.float32)).to(dev, torch.float16)
x_T = torch.from_numpy(rs(42).randn(1, 4, 64, 64).astype(np.float32)).to(dev)
steps = int(sys.argv[3])
sync = int(sys.argv[4])
orig = net.forward_nhwc
calls = [0]
def checked(x, t, context, **kw):
    y = orig(x, t, context, **kw)
    calls[0] += 1
    if sync:
        torch.cuda.synchronize()
        if not torch.isfinite(y).all():
            print("non-finite at call", calls[0]); sys.exit(1)
    return y
net.forward_nhwc = checked
for rep in range(2):
    out = pipe(c=c, uc=uc, x_T=x_T, H=512, W=512, steps=steps, scale=9.0, eta=0.0, decode=False, batch_size=1)
    print(sys.argv[1:], "rep", rep, "finite", bool(torch.isfinite(out).all()), "absmax", float(out.float().nan_to_num(0,0,0).abs().max()), flush=True)
