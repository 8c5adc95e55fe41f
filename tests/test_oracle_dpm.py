"""CPU tests that pin the DPM-Solver++ restatement (oracle/dpm_solver.py) and the product's host-side coefficient plan
(minddiffusion_amd/ldm/models/diffusion/dpm_solver/dpm_solver.py) against each other and against known identities."""
import numpy as np
import torch

from oracle import dpm_solver as OD
from oracle import ldm as O


def _schedule():
    return O.register_schedule()          # SD linear schedule, 1000 steps


def test_discrete_schedule_known_answers():
    s = _schedule()
    ns = OD.NoiseScheduleVP("discrete", alphas_cumprod=s["alphas_cumprod"])
    ac = np.asarray(s["alphas_cumprod"], np.float64)
    # on the grid t_k = (k+1)/1000 the interpolation returns the table: log alpha_t = 0.5 log alpha-bar_k
    for k in (0, 1, 499, 998, 999):
        t = torch.tensor([(k + 1) / 1000.0], dtype=torch.float64)
        assert abs(float(ns.marginal_log_mean_coeff(t)) - 0.5 * np.log(ac[k])) < 1e-12
        assert abs(float(ns.marginal_alpha(t)) ** 2 + float(ns.marginal_std(t)) ** 2 - 1.0) < 1e-12
    # between grid points: linear in t
    t = torch.tensor([500.5 / 1000.0], dtype=torch.float64)
    mid = 0.25 * (np.log(ac[499]) + np.log(ac[500]))
    assert abs(float(ns.marginal_log_mean_coeff(t)) - mid) < 1e-12
    # lambda is strictly decreasing in t and inverse_lambda inverts it
    tt = torch.linspace(0.001, 1.0, 57, dtype=torch.float64)
    lam = ns.marginal_lambda(tt)
    assert bool((lam[1:] < lam[:-1]).all())
    assert float((ns.inverse_lambda(lam) - tt).abs().max()) < 1e-9


def test_product_schedule_and_plan_equal_oracle():
    from minddiffusion_amd.ldm.models.diffusion.dpm_solver.dpm_solver import NoiseScheduleVP, multistep_2m_plan
    s = _schedule()
    ac = np.asarray(s["alphas_cumprod"], np.float64)
    pn, on = NoiseScheduleVP("discrete", alphas_cumprod=ac), OD.NoiseScheduleVP("discrete", alphas_cumprod=ac)
    tt = np.concatenate([np.linspace(0.001, 1.0, 211), [0.0005, 1.0004]])    # incl. the extrapolated ends
    to = torch.tensor(tt, dtype=torch.float64)
    for name in ("marginal_log_mean_coeff", "marginal_alpha", "marginal_std", "marginal_lambda"):
        np.testing.assert_allclose(getattr(pn, name)(tt), getattr(on, name)(to).numpy(), rtol=1e-12, atol=1e-13)
    lam = on.marginal_lambda(to[:-2])
    np.testing.assert_allclose(pn.inverse_lambda(lam.numpy()), on.inverse_lambda(lam).numpy(), rtol=1e-10, atol=1e-12)

    # the closed-form plan (x_next = A x + c0 x0_k + c1 x0_{k-1}) must reproduce the tensor-level solver exactly on a
    # toy noise model (any deterministic function of (x, t) will do)
    def toy_eps(x, t_input, cond):
        return torch.tanh(0.3 * x + 0.001 * t_input.reshape(-1, 1, 1, 1).to(x.dtype)) + 0.1 * cond

    rng = np.random.RandomState(0)
    x_T = rng.randn(3, 4, 5, 5)
    cond = torch.tensor(rng.randn(3, 1, 1, 1))
    for S in (2, 5, 10, 14, 15, 20):
        model_fn = OD.model_wrapper(toy_eps, on, cond, None, 1.0)
        solver = OD.DPM_Solver(model_fn, on, predict_x0=True)
        ref = solver.sample(torch.tensor(x_T), steps=S)
        assert solver.nfe == S                                    # S model evaluations for S steps
        plan = multistep_2m_plan(pn, S)
        assert len(plan) == S and plan[0]["order"] == 1
        assert plan[-1]["order"] == (1 if S < 15 else 2)          # lower_order_final only below 15 steps
        x, prev = torch.tensor(x_T), None
        for p in plan:
            e = toy_eps(x, torch.full((3,), p["t_input"], dtype=torch.float64).to(torch.float32), cond)   # UNet takes fp32 t
            x0 = (x - p["sigma"] * e) / p["alpha"]
            x = p["A"] * x + p["c0"] * x0 + (p["c1"] * prev if p["c1"] != 0.0 else 0.0)
            prev = x0
        np.testing.assert_allclose(x.numpy(), ref.numpy(), rtol=1e-9, atol=1e-10)
        assert abs(plan[0]["t_input"] - 999.0) < 1e-9 and abs(plan[-1]["t_next"] - 0.001) < 1e-12


def test_first_order_update_is_ddim():
    """DPM-Solver-1 in data-prediction form == the DDIM step (eta = 0) between the same two grid times
    ('DDIM(eta=0) == PLMS update with e'=e_t == DPM-Solver-1')."""
    s = _schedule()
    ac = np.asarray(s["alphas_cumprod"], np.float64)
    ns = OD.NoiseScheduleVP("discrete", alphas_cumprod=ac)
    rng = np.random.RandomState(1)
    x = torch.tensor(rng.randn(2, 4, 3, 3))
    eps = torch.tensor(rng.randn(2, 4, 3, 3))
    ks, kt = 800, 780                                              # table indices: t = (k+1)/1000
    s_t = torch.full((2,), (ks + 1) / 1000.0, dtype=torch.float64)
    t_t = torch.full((2,), (kt + 1) / 1000.0, dtype=torch.float64)
    solver = OD.DPM_Solver(lambda xx, tt: eps, ns, predict_x0=True)
    x0 = (x - np.sqrt(1 - ac[ks]) * eps) / np.sqrt(ac[ks])
    got = solver.dpm_solver_first_update(x, s_t, t_t, model_s=x0)
    ddim = np.sqrt(ac[kt]) * x0 + np.sqrt(1 - ac[kt]) * eps         # plms.py:218-226 with sigma = 0
    np.testing.assert_allclose(got.numpy(), ddim.numpy(), rtol=1e-10, atol=1e-12)
