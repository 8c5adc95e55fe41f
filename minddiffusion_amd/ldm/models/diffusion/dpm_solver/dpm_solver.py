"""Host-side math of DPM-Solver++ as the reference wires it (ldm/models/diffusion/dpm_solver/dpm_solver.py):
``NoiseScheduleVP('discrete')`` (:79-164), the ``time_uniform`` grid (:414-415) and the multistep order-2 update in
data-prediction form (``multistep_dpm_solver_second_update`` :742-797, first-order start / tail :488-532).

Everything here is float64 numpy on S+1 scalars; the per-element work of a step is ONE launch of the fused
``mdx_sampler_step_f32`` kernel (see sampler.py), so there is no tensor math in this file.
"""
import numpy as np


class NoiseScheduleVP:
    """dpm_solver.py:14-164, 'discrete' schedule only (the only one sampler.py:72 constructs)."""

    def __init__(self, schedule="discrete", betas=None, alphas_cumprod=None):
        if schedule != "discrete":
            raise NotImplementedError("the reference's sampler only builds NoiseScheduleVP('discrete', alphas_cumprod=...)")
        if betas is not None:
            log_alphas = 0.5 * np.cumsum(np.log(1.0 - np.asarray(betas, np.float64)))
        else:
            assert alphas_cumprod is not None
            log_alphas = 0.5 * np.log(np.asarray(alphas_cumprod, np.float64))
        self.schedule = schedule
        self.total_N = len(log_alphas)
        self.T = 1.0
        self.t_array = np.linspace(0.0, 1.0, self.total_N + 1)[1:]       # t_k = (k + 1) / N
        self.log_alpha_array = log_alphas

    @staticmethod
    def _interp(x, xp, yp):
        """interpolate_fn :1126-1171: piecewise linear through (xp, yp), the outermost segments extended linearly."""
        x = np.asarray(x, np.float64)
        idx = np.clip(np.searchsorted(xp, x, side="left"), 1, len(xp) - 1)
        x0, x1, y0, y1 = xp[idx - 1], xp[idx], yp[idx - 1], yp[idx]
        return y0 + (x - x0) * (y1 - y0) / (x1 - x0)

    def marginal_log_mean_coeff(self, t):
        return self._interp(t, self.t_array, self.log_alpha_array)

    def marginal_alpha(self, t):
        return np.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return np.sqrt(1.0 - np.exp(2.0 * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t):
        lm = self.marginal_log_mean_coeff(t)
        return lm - 0.5 * np.log(1.0 - np.exp(2.0 * lm))

    def inverse_lambda(self, lamb):
        log_alpha = -0.5 * np.logaddexp(0.0, -2.0 * np.asarray(lamb, np.float64))
        return self._interp(log_alpha, self.log_alpha_array[::-1], self.t_array[::-1])

    def model_input_time(self, t_continuous):
        """model_wrapper.get_model_input_time :256-265: continuous t in [1/N, 1] -> the UNet's (fractional) timestep."""
        return (np.asarray(t_continuous, np.float64) - 1.0 / self.total_N) * 1000.0


def time_uniform_steps(ns, steps, t_start=None, t_end=None):
    """DPM_Solver.get_time_steps(skip_type='time_uniform') :414-415 with sample()'s defaults :1040-1041."""
    t_0 = 1.0 / ns.total_N if t_end is None else t_end
    t_T = ns.T if t_start is None else t_start
    return np.linspace(t_T, t_0, steps + 1)


def multistep_2m_plan(ns, steps, order=2, lower_order_final=True):
    """The S updates of DPM_Solver.sample(method='multistep', order=2, predict_x0=True) :1044-1075 as scalars.

    Step k (k = 0 .. S-1) evaluates the model at timesteps[k] (x0_k = (x - sigma_k eps) / alpha_k) and moves x from
    timesteps[k] to timesteps[k+1]:
        x_next = A * x + c0 * x0_k + c1 * x0_{k-1}
    first order  (:488-511):  A = sigma_t / sigma_s, c0 = -alpha_t * expm1(-h),                     c1 = 0
    second order (:742-773):  A = sigma_t / sigma_s, c0 = -alpha_t (e^{-h} - 1) (1 + 1 / (2 r0)),   c1 = alpha_t (e^{-h} - 1) / (2 r0)
    with h = lambda_t - lambda_s, r0 = (lambda_s - lambda_{s-1}) / h.  S model evaluations in total (the last
    update's target needs none, :1073-1075)."""
    assert order in (1, 2) and steps >= order
    ts = time_uniform_steps(ns, steps)
    lam, alpha, sigma = ns.marginal_lambda(ts), ns.marginal_alpha(ts), ns.marginal_std(ts)
    plan = []
    for k in range(steps):
        step = k + 1                                    # index of the target time in `ts`
        if k == 0:
            step_order = 1                              # init by the lower-order solver (:1050-1056)
        elif lower_order_final and steps < 15:
            step_order = min(order, steps + 1 - step)   # :1060-1063
        else:
            step_order = order
        h = lam[k + 1] - lam[k]
        A = sigma[k + 1] / sigma[k]
        if step_order == 1:
            c0, c1 = -alpha[k + 1] * np.expm1(-h), 0.0
        else:
            r0 = (lam[k] - lam[k - 1]) / h
            phi = alpha[k + 1] * (np.exp(-h) - 1.0)
            c0, c1 = -phi * (1.0 + 0.5 / r0), 0.5 * phi / r0
        plan.append(dict(t=ts[k], t_next=ts[k + 1], t_input=float(ns.model_input_time(ts[k])), alpha=alpha[k],
                         sigma=sigma[k], A=A, c0=c0, c1=c1, order=step_order))
    return plan
