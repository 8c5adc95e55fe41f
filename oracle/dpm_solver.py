"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- PARITY UNPINNED by the reference (no tests / fixtures; MindSpore
cannot run here).

fp32/float64 PyTorch-CPU restatement of DPM-Solver++ as the reference wires it:
    DPMSolverSampler.sample     /root/reference/vision/stablediffusionv2/ldm/models/diffusion/dpm_solver/sampler.py:30-92
    NoiseScheduleVP('discrete') .../dpm_solver/dpm_solver.py:79-164
    model_wrapper (CFG)         .../dpm_solver.py:180-335
    DPM_Solver.sample(multistep), dpm_solver_first_update, multistep_dpm_solver_second_update
                                .../dpm_solver.py:488-532, 742-797, 1040-1075
    interpolate_fn              .../dpm_solver.py:1126-1171
The tensor-level structure of the reference is kept (model_prev_list / t_prev_list rotation, per-sample time vectors,
sort-based interpolation) so that it is an independent check of the product's closed-form coefficient plan.  Oracle mode
is fp32 for images and float64 for the schedule; the reference's fp16 casts (:100-103, :415, sampler.py:88) are not kept.
"""
import torch


def interpolate_fn(x, xp, yp):
    """dpm_solver.py:1126-1171, same sort / gather construction.  x [N, C], xp / yp [C, K]."""
    N, K = x.shape[0], xp.shape[1]
    all_x = torch.cat([x.unsqueeze(2), xp.unsqueeze(0).repeat((N, 1, 1))], dim=2)
    sorted_all_x, x_indices = torch.sort(all_x, dim=2)
    x_idx = torch.argmin(x_indices, dim=2)
    cand_start_idx = x_idx - 1
    start_idx = torch.where(torch.eq(x_idx, 0), torch.tensor(1),
                            torch.where(torch.eq(x_idx, K), torch.tensor(K - 2), cand_start_idx))
    end_idx = torch.where(torch.eq(start_idx, cand_start_idx), start_idx + 2, start_idx + 1)
    start_x = torch.gather(sorted_all_x, dim=2, index=start_idx.unsqueeze(2)).squeeze(2)
    end_x = torch.gather(sorted_all_x, dim=2, index=end_idx.unsqueeze(2)).squeeze(2)
    start_idx2 = torch.where(torch.eq(x_idx, 0), torch.tensor(0),
                             torch.where(torch.eq(x_idx, K), torch.tensor(K - 2), cand_start_idx))
    y_positions_expanded = yp.unsqueeze(0).expand(N, -1, -1)
    start_y = torch.gather(y_positions_expanded, dim=2, index=start_idx2.unsqueeze(2)).squeeze(2)
    end_y = torch.gather(y_positions_expanded, dim=2, index=(start_idx2 + 1).unsqueeze(2)).squeeze(2)
    return start_y + (x - start_x) * (end_y - start_y) / (end_x - start_x)


class NoiseScheduleVP:
    def __init__(self, schedule="discrete", betas=None, alphas_cumprod=None):
        assert schedule == "discrete"
        if betas is not None:
            log_alphas = 0.5 * torch.log(1 - torch.as_tensor(betas, dtype=torch.float64)).cumsum(dim=0)
        else:
            log_alphas = 0.5 * torch.log(torch.as_tensor(alphas_cumprod, dtype=torch.float64))
        self.schedule = schedule
        self.total_N = len(log_alphas)
        self.T = 1.0
        self.t_array = torch.linspace(0.0, 1.0, self.total_N + 1, dtype=torch.float64)[1:].reshape((1, -1))
        self.log_alpha_array = log_alphas.reshape((1, -1))

    def marginal_log_mean_coeff(self, t):
        return interpolate_fn(t.reshape((-1, 1)), self.t_array, self.log_alpha_array).reshape((-1,))

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t):
        log_mean_coeff = self.marginal_log_mean_coeff(t)
        log_std = 0.5 * torch.log(1.0 - torch.exp(2.0 * log_mean_coeff))
        return log_mean_coeff - log_std

    def inverse_lambda(self, lamb):
        log_alpha = -0.5 * torch.logaddexp(torch.zeros((1,), dtype=lamb.dtype), -2.0 * lamb)
        t = interpolate_fn(log_alpha.reshape((-1, 1)), torch.flip(self.log_alpha_array, [1]), torch.flip(self.t_array, [1]))
        return t.reshape((-1,))


def _expand(v, dims):
    return v[(...,) + (None,) * (dims - 1)]


def model_wrapper(model, noise_schedule, condition, unconditional_condition, guidance_scale):
    """dpm_solver.py:180-335 for model_type='noise', guidance_type='classifier-free' (sampler.py:74-82).
    `model(x, t_input, cond)` is LatentDiffusion.apply_model."""

    def get_model_input_time(t_continuous):
        return (t_continuous - 1.0 / noise_schedule.total_N) * 1000.0

    def noise_pred_fn(x, t_continuous, cond):
        return model(x, get_model_input_time(t_continuous).to(torch.float32), cond)

    def model_fn(x, t_continuous):
        if t_continuous.reshape((-1,)).shape[0] == 1:
            t_continuous = t_continuous.expand((x.shape[0],))
        if guidance_scale == 1.0 or unconditional_condition is None:
            return noise_pred_fn(x, t_continuous, condition)
        x_in = torch.cat([x] * 2)
        t_in = torch.cat([t_continuous] * 2)
        c_in = torch.cat([unconditional_condition, condition])
        noise_uncond, noise = noise_pred_fn(x_in, t_in, c_in).chunk(2)
        return noise_uncond + guidance_scale * (noise - noise_uncond)

    return model_fn


class DPM_Solver:
    def __init__(self, model_fn, noise_schedule, predict_x0=True):
        self.model = model_fn
        self.noise_schedule = noise_schedule
        self.predict_x0 = predict_x0
        self.nfe = 0

    def model_fn(self, x, t):          # data_prediction_fn :372-389 (thresholding=False) / noise_prediction_fn
        self.nfe += 1
        noise = self.model(x, t)
        if not self.predict_x0:
            return noise
        ns = self.noise_schedule
        alpha_t, sigma_t = ns.marginal_alpha(t), ns.marginal_std(t)
        return (x - _expand(sigma_t, x.ndim).to(x.dtype) * noise) / _expand(alpha_t, x.ndim).to(x.dtype)

    def dpm_solver_first_update(self, x, s, t, model_s):          # :488-532
        ns = self.noise_schedule
        lambda_s, lambda_t = ns.marginal_lambda(s), ns.marginal_lambda(t)
        h = lambda_t - lambda_s
        log_alpha_s, log_alpha_t = ns.marginal_log_mean_coeff(s), ns.marginal_log_mean_coeff(t)
        sigma_s, sigma_t = ns.marginal_std(s), ns.marginal_std(t)
        alpha_t = torch.exp(log_alpha_t)
        d = x.ndim
        if self.predict_x0:
            phi_1 = torch.expm1(-h)
            return _expand(sigma_t / sigma_s, d).to(x.dtype) * x - _expand(alpha_t * phi_1, d).to(x.dtype) * model_s
        phi_1 = torch.expm1(h)
        return (_expand(torch.exp(log_alpha_t - log_alpha_s), d).to(x.dtype) * x
                - _expand(sigma_t * phi_1, d).to(x.dtype) * model_s)

    def multistep_dpm_solver_second_update(self, x, model_prev_list, t_prev_list, t):   # :742-797, 'dpm_solver' type
        ns = self.noise_schedule
        model_prev_1, model_prev_0 = model_prev_list
        t_prev_1, t_prev_0 = t_prev_list
        lambda_prev_1, lambda_prev_0, lambda_t = ns.marginal_lambda(t_prev_1), ns.marginal_lambda(t_prev_0), ns.marginal_lambda(t)
        log_alpha_prev_0, log_alpha_t = ns.marginal_log_mean_coeff(t_prev_0), ns.marginal_log_mean_coeff(t)
        sigma_prev_0, sigma_t = ns.marginal_std(t_prev_0), ns.marginal_std(t)
        alpha_t = torch.exp(log_alpha_t)
        h_0 = lambda_prev_0 - lambda_prev_1
        h = lambda_t - lambda_prev_0
        r0 = h_0 / h
        d = x.ndim
        D1_0 = _expand(1.0 / r0, d).to(x.dtype) * (model_prev_0 - model_prev_1)
        if self.predict_x0:
            return (_expand(sigma_t / sigma_prev_0, d).to(x.dtype) * x
                    - _expand(alpha_t * (torch.exp(-h) - 1.0), d).to(x.dtype) * model_prev_0
                    - 0.5 * _expand(alpha_t * (torch.exp(-h) - 1.0), d).to(x.dtype) * D1_0)
        return (_expand(torch.exp(log_alpha_t - log_alpha_prev_0), d).to(x.dtype) * x
                - _expand(sigma_t * (torch.exp(h) - 1.0), d).to(x.dtype) * model_prev_0
                - 0.5 * _expand(sigma_t * (torch.exp(h) - 1.0), d).to(x.dtype) * D1_0)

    def multistep_dpm_solver_update(self, x, model_prev_list, t_prev_list, t, order):   # :877-901
        if order == 1:
            return self.dpm_solver_first_update(x, t_prev_list[-1], t, model_s=model_prev_list[-1])
        assert order == 2
        return self.multistep_dpm_solver_second_update(x, model_prev_list, t_prev_list, t)

    def sample(self, x, steps, order=2, lower_order_final=True):
        """:1040-1075 with skip_type='time_uniform', method='multistep'."""
        ns = self.noise_schedule
        t_0, t_T = 1.0 / ns.total_N, ns.T
        assert steps >= order
        timesteps = torch.linspace(t_T, t_0, steps + 1, dtype=torch.float64)     # get_time_steps :414-415
        vec_t = timesteps[0].expand((x.shape[0],))
        model_prev_list = [self.model_fn(x, vec_t)]
        t_prev_list = [vec_t]
        for init_order in range(1, order):
            vec_t = timesteps[init_order].expand((x.shape[0],))
            x = self.multistep_dpm_solver_update(x, model_prev_list, t_prev_list, vec_t, init_order)
            model_prev_list.append(self.model_fn(x, vec_t))
            t_prev_list.append(vec_t)
        for step in range(order, steps + 1):
            vec_t = timesteps[step].expand((x.shape[0],))
            step_order = min(order, steps + 1 - step) if (lower_order_final and steps < 15) else order
            x = self.multistep_dpm_solver_update(x, model_prev_list, t_prev_list, vec_t, step_order)
            for i in range(order - 1):
                t_prev_list[i] = t_prev_list[i + 1]
                model_prev_list[i] = model_prev_list[i + 1]
            t_prev_list[-1] = vec_t
            if step < steps:
                model_prev_list[-1] = self.model_fn(x, vec_t)
        return x


def sample(model, S, batch_size, shape, conditioning, x_T, unconditional_guidance_scale=1.0,
           unconditional_conditioning=None):
    """DPMSolverSampler.sample sampler.py:30-92.  `model`: an object with alphas_cumprod and apply_model(x, t, cond)
    (oracle.ldm.ModelOracle).  Returns (samples, solver) -- solver.nfe is the number of model evaluations."""
    img = torch.as_tensor(x_T, dtype=torch.float32)
    assert tuple(img.shape) == (batch_size,) + tuple(shape)
    cond = torch.as_tensor(conditioning, dtype=torch.float32)
    uc = None if unconditional_conditioning is None else torch.as_tensor(unconditional_conditioning, dtype=torch.float32)
    ns = NoiseScheduleVP("discrete", alphas_cumprod=model.alphas_cumprod)
    model_fn = model_wrapper(lambda x, t, c: model.apply_model(x, t, c), ns, cond, uc, float(unconditional_guidance_scale))
    solver = DPM_Solver(model_fn, ns, predict_x0=True)
    x = solver.sample(img, steps=S, order=2, lower_order_final=True)
    return x, solver
