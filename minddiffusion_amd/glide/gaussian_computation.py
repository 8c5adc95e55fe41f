"""Beta schedules and timestep respacing -- host-side mirror of the reference's
Taichu-GLIDE/model/glide_text2im/gaussian_computation.py (get_named_beta_schedule :20, betas_for_alpha_bar :48,
get_beta_schedule :62 (the deprecated quad / linear / warmup10 / warmup50 / const / jsd forms), space_timesteps :109 incl. the
"ddimN" stride form :129-135, alpha_calculator :166).  Pure numpy float64, as in the reference; pinned against goldens
generated from the reference module itself (tests/golden/glide_schedule.npz)."""
import math

import numpy as np


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    betas = []
    for i in range(num_diffusion_timesteps):
        t1, t2 = i / num_diffusion_timesteps, (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return np.array(betas)


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "squaredcos_cap_v2":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps):
    """The older schedule family (gaussian_computation.py:62-106), float64 like the reference."""
    n = num_diffusion_timesteps

    def ramp_then_flat(frac):      # linear warm-up over the first int(n * frac) steps, beta_end afterwards
        out = np.full(n, beta_end, dtype=np.float64)
        k = int(n * frac)
        out[:k] = np.linspace(beta_start, beta_end, k, dtype=np.float64)
        return out
    forms = {
        "quad": lambda: np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float64) ** 2,
        "linear": lambda: np.linspace(beta_start, beta_end, n, dtype=np.float64),
        "warmup10": lambda: ramp_then_flat(0.1),
        "warmup50": lambda: ramp_then_flat(0.5),
        "const": lambda: np.full(n, beta_end, dtype=np.float64),
        "jsd": lambda: 1.0 / np.linspace(n, 1, n, dtype=np.float64),      # 1/T, 1/(T-1), ..., 1
    }
    if beta_schedule not in forms:
        raise NotImplementedError(beta_schedule)
    betas = forms[beta_schedule]()
    assert betas.shape == (n,)
    return betas


def space_timesteps(num_timesteps, section_counts):
    """Evenly strided subsets per section ("60", "100,50", ...), the DDIM-paper stride "ddimN" (the first integer stride that
    yields exactly N steps) and the hand-tuned "fast27" set."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                picked = range(0, num_timesteps, stride)
                if len(picked) == want:
                    return set(picked)
            raise ValueError(f"cannot create exactly {want} steps with an integer stride over {num_timesteps}")
        if section_counts == "fast27":
            steps = space_timesteps(num_timesteps, "10,10,3,2,2")
            steps.remove(num_timesteps - 1)
            steps.add(num_timesteps - 3)
            return steps
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start_idx, all_steps = 0, []
    for i, section_count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < section_count:
            raise ValueError(f"cannot divide section of {size} steps into {section_count}")
        frac_stride = 1 if section_count <= 1 else (size - 1) / (section_count - 1)
        cur_idx = 0.0
        for _ in range(section_count):
            all_steps.append(start_idx + round(cur_idx))
            cur_idx += frac_stride
        start_idx += size
    return set(all_steps)


def alpha_calculator(betas):
    betas = np.array(betas, dtype=np.float64)
    assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
    return np.cumprod(1.0 - betas, axis=0)
