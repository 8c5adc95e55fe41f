"""debug: UNet batch 8 vs batch 2 row consistency, Wukong 4-ch and 9-ch"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from minddiffusion_amd.configs import WUKONG_INPAINT_UNET, WUKONG_UNET
from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
from minddiffusion_amd.weights import synthetic_unet_params_device
from minddiffusion_amd import ops
DEV = "cuda:0"
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for name, cfg in (("wukong9", WUKONG_INPAINT_UNET), ("wukong4", WUKONG_UNET)):
    net = UNetModel(**dict(cfg)); net.use_graph = False
    net.load_state_dict(synthetic_unet_params_device(net.parameter_shapes(), seed=0, device=DEV))
    C = cfg["in_channels"]
    rng = np.random.RandomState(0)
    x = torch.tensor(rng.randn(8, C, 64, 64).astype(np.float32), device=DEV)
    ctx = torch.tensor(rng.randn(8, 77, 768).astype(np.float32), device=DEV)
    t = torch.full((8,), 500.0, device=DEV)
    full = net(x, t, ctx).clone()
    for B in (2, 4):
        for r0 in (0, 8 - B):
            part = net(x[r0:r0 + B], t[:B], ctx[r0:r0 + B])
            print(name, f"B8 rows {r0}:{r0 + B} vs B{B}", rel(full[r0:r0 + B], part))
    # where does it go wrong: compare per-op outputs of plan B=8 and plan B=2 on rows 0:2? (first conv output)
    P8, P2 = net._plans[(8, 64, 64)], net._plans[(2, 64, 64)]
    print(name, "ops", len(P8.main), len(P2.main))
    for m in P8.meta[:12]:
        print("   ", m["kind"], m["info"], m.get("launches"))
