"""Generate tests/golden/glide_schedule.npz by IMPORTING the reference's MindSpore-free
gaussian_computation.py (run in the build container only; /root/reference does not travel).

    python tests/golden/make_glide_schedule_golden.py
"""
import importlib.util
import os

import numpy as np

REF = "/root/reference/vision/Taichu-GLIDE/model/glide_text2im/gaussian_computation.py"
spec = importlib.util.spec_from_file_location("ref_gaussian_computation", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

out = {}
out["cosine_betas_1000"] = np.asarray(mod.get_named_beta_schedule("squaredcos_cap_v2", 1000), dtype=np.float64)
out["linear_betas_1000"] = np.asarray(mod.get_named_beta_schedule("linear", 1000), dtype=np.float64)
out["space_60"] = np.asarray(sorted(mod.space_timesteps(1000, "60")), dtype=np.int64)
out["space_fast27"] = np.asarray(sorted(mod.space_timesteps(1000, "fast27")), dtype=np.int64)
out["space_100_50"] = np.asarray(sorted(mod.space_timesteps(1000, "100,50")), dtype=np.int64)
out["space_ddim25"] = np.asarray(sorted(mod.space_timesteps(1000, "ddim25")), dtype=np.int64)
out["space_ddim50"] = np.asarray(sorted(mod.space_timesteps(1000, "ddim50")), dtype=np.int64)
out["space_ddim100_of_4000"] = np.asarray(sorted(mod.space_timesteps(4000, "ddim100")), dtype=np.int64)
for name in ("quad", "linear", "warmup10", "warmup50", "const", "jsd"):
    out[f"legacy_{name}_1000"] = np.asarray(mod.get_beta_schedule(name, beta_start=0.0001, beta_end=0.02,
                                                                   num_diffusion_timesteps=1000), dtype=np.float64)
    out[f"legacy_{name}_37"] = np.asarray(mod.get_beta_schedule(name, beta_start=0.001, beta_end=0.05,
                                                                 num_diffusion_timesteps=37), dtype=np.float64)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "glide_schedule.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, {k: v.shape for k, v in out.items()})
