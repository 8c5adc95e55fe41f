"""debug: per-row consistency of UNet batch B vs B=1 runs"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from minddiffusion_amd.configs import SD2_UNET, WUKONG_UNET
from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
from minddiffusion_amd.weights import synthetic_unet_params_device
DEV = "cuda:0"
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
which = sys.argv[1]
cfgs = {"wukong": (WUKONG_UNET, 768), "sd2": (SD2_UNET, 1024),
        "wukong_l1": (dict(WUKONG_UNET, channel_mult=[1], attention_resolutions=[1]), 768),
        "wukong_l2": (dict(WUKONG_UNET, channel_mult=[1, 2], attention_resolutions=[1, 2]), 768),
        "wukong_noattn": (dict(WUKONG_UNET, attention_resolutions=[]), 768)}
cfg, cd = cfgs[which]
net = UNetModel(**dict(cfg)); net.use_graph = False
net.load_state_dict(synthetic_unet_params_device(net.parameter_shapes(), seed=0, device=DEV))
rng = np.random.RandomState(0)
for B in [int(v) for v in sys.argv[2].split(",")]:
    x = torch.tensor(rng.randn(B, 4, 64, 64).astype(np.float32), device=DEV)
    ctx = torch.tensor(rng.randn(B, 77, cd).astype(np.float32), device=DEV)
    t = torch.full((B,), 500.0, device=DEV)
    full = net(x, t, ctx).clone()
    errs = []
    for r in range(B):
        one = net(x[r:r + 1].clone(), t[:1], ctx[r:r + 1].clone())
        errs.append(rel(full[r:r + 1], one))
    print(which, "B", B, " ".join(f"{e:.1e}" for e in errs), flush=True)
