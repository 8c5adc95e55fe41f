"""Schedule helpers -- host-side mirror of the reference's ldm/modules/diffusionmodules/util.py
(make_beta_schedule :172-185, make_ddim_timesteps :134-148, make_ddim_sampling_parameters :151-162).
numpy fp32, exactly as the reference computes them on the host (betas.asnumpy(), np.cumprod)."""
import numpy as np


def make_beta_schedule(schedule="linear", n_timestep=1000, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule != "linear":
        raise ValueError(f"schedule '{schedule}' unknown.")
    start = np.float32(linear_start ** 0.5)
    stop = np.float32(linear_end ** 0.5)
    return (np.linspace(start, stop, n_timestep, dtype=np.float32) ** 2).astype(np.float32)


def make_ddim_timesteps(ddim_discr_method="uniform", num_ddim_timesteps=50, num_ddpm_timesteps=1000, verbose=False):
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)), dtype=np.int64)
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = ddim_timesteps + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta=0.0, verbose=False):
    alphacums = np.asarray(alphacums, dtype=np.float32)
    if int(np.max(ddim_timesteps)) >= alphacums.shape[0]:
        raise IndexError(f"ddim timestep {int(np.max(ddim_timesteps))} out of range for {alphacums.shape[0]} "
                         "training steps (choose S dividing num_timesteps)")
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.concatenate([alphacums[:1], alphacums[ddim_timesteps[:-1]]]).astype(np.float32)
    sigmas = (np.float32(eta) * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))).astype(np.float32)
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the following sigma_t schedule "
              f"for ddim sampler {sigmas}")
    return sigmas, alphas, alphas_prev
