"""debug: full-size inpainting trajectory vs fixture"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from make_trajectory_goldens import inputs_inpaint
from oracle import ldm as O
from minddiffusion_amd.configs import WUKONG_INPAINT_UNET
from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentInpaintDiffusion
from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
DEV = "cuda:0"
z = np.load(os.path.join(ROOT, "tests", "golden", "traj_inpaint_wukong_plms30.npz"))
inp = inputs_inpaint()
ocfg = dict(O.WUKONG_UNET, in_channels=9)
params = O.init_params(ocfg, seed=inp["seed"])
net = UNetModel(**dict(WUKONG_INPAINT_UNET)); net.use_graph = True; net.load_state_dict(params)
model = LatentInpaintDiffusion(unet_config=net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
dev = lambda a: torch.tensor(a, device=DEV)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
ref = torch.tensor(z["final"].astype(np.float32))
def run(B, S, graph=True, temb=True):
    net.use_graph = graph
    os.environ["MDX_SAMPLER_TEMB_TABLE"] = "1" if temb else "0"
    got, inter = PLMSSampler(model).sample(S, B, (4, 64, 64), conditioning={"c_concat": dev(inp["c_cat"][:B]), "c_crossattn": dev(inp["c"][:B])},
                                   x_T=dev(inp["x_T"][:B]), unconditional_guidance_scale=inp["scale"],
                                   unconditional_conditioning={"c_concat": dev(inp["c_cat"][:B]), "c_crossattn": dev(inp["uc"][:B])}, verbose=False, log_every_t=1)
    return got.cpu(), inter
g4, i4 = run(4, 30)
g1, i1 = run(1, 30)
print("B4 vs fixture", rel(g4[:1], ref), "B1 vs fixture", rel(g1, ref), "B4 vs B1", rel(g4[:1], g1))
g1e, _ = run(1, 30, graph=False, temb=False)
print("B1 eager/no-table vs B1", rel(g1e, g1))
# per-step divergence of B4 vs B1
for k in range(0, len(i4["x_inter"]), 3):
    print(k, rel(i4["x_inter"][k][:1].cpu(), i1["x_inter"][k].cpu()), float(i1["x_inter"][k].abs().max()))
# oracle: first 2 PLMS steps (S=30 grid) on CPU for image 0
torch.set_num_threads(min(96, os.cpu_count() or 8))
om = O.ModelOracle(O.UNetOracle(ocfg, params), conditioning_key="hybrid")
ro, io = O.sample(om, 30, 1, (4, 64, 64), {"c_concat": inp["c_cat"][:1], "c_crossattn": inp["c"][:1]}, inp["x_T"][:1], "plms",
                  unconditional_guidance_scale=inp["scale"], unconditional_conditioning={"c_concat": inp["c_cat"][:1], "c_crossattn": inp["uc"][:1]},
                  timesteps=4, log_every_t=1)
print("oracle prefix steps:", len(io["x_inter"]))
for k in range(len(io["x_inter"])):
    print("step", k, "gpu B1 vs oracle", rel(i1["x_inter"][k].cpu(), io["x_inter"][k]), float(io["x_inter"][k].abs().max()))
