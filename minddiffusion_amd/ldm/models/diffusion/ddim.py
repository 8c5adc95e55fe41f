"""DDIMSampler -- the LDM DDIM sampler the north star names.  The reference has NO ``DDIMSampler`` class
(SURVEY.md 0.4): this is its single-step update ``get_x_prev_and_pred_x0``
(ldm/models/diffusion/plms.py:210-228) applied with e' = e_t every step -- S UNet calls, no Adams-Bashforth
history -- which is also what ``dpm_solver_first_update`` documents itself as
(ldm/models/diffusion/dpm_solver/dpm_solver.py:488-532).  Same ``sample(...)`` signature as PLMSSampler;
``eta`` may be non-zero (sigma_t * N(0,1) * temperature is added, plms.py:223-226)."""
from .plms import _SamplerBase


class DDIMSampler(_SamplerBase):
    multistep = False

    def ddim_sampling(self, *args, **kwargs):
        return self.plms_sampling(*args, **kwargs)
