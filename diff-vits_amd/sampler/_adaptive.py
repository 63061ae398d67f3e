"""`DPM_Solver.sample(method='adaptive')`: the adaptive step-size solver on the singlestep updates (reference
sampler/dpm_solver.py:906-1010 `dpm_solver_adaptive`, with `dpm_solver_first_update` :547-592,
`singlestep_dpm_solver_second_update` :594-676 and `singlestep_dpm_solver_third_update` :678-794).

The step sizes depend on the data (an error estimate between the lower- and the higher-order update decides whether a step
is accepted and how long the next one is), so this loop cannot be compiled ahead like the multistep / singlestep plans: it
runs from Python - torch ops on the state, one model evaluation per stage (natively through `NativeUNetModel.__call__` when the
model is this package's UNet) and one host decision per step.
"""
import torch


def _prediction_fn(solver):
    """(x, t[1]) -> what the algorithm's updates consume: the data prediction ('dpmsolver++', with correcting_x0_fn applied,
    dpm_solver.py:433-445) or the noise prediction ('dpmsolver')."""
    ns, model_fn = solver.noise_schedule, solver.model_fn
    data = solver.algorithm_type == "dpmsolver++"

    def bc(v, x):
        return v.to(x).reshape((-1,) + (1,) * (x.dim() - 1))

    def fn(x, t):
        tb = t.expand(x.shape[0])
        # Always through the wrapper's noise prediction and back, exactly as the reference's data_prediction_fn does
        # (dpm_solver.py:433-445) - also for an x_start network, where x0 -> noise -> x0 is the identity only up to rounding:
        # the accept test `E <= 1` below is data-dependent, so a 1e-7 difference in a prediction changes accept / reject
        # decisions and with them the whole trajectory (round 4 used the network's x0 directly: 30 evaluations where the
        # reference takes 27, result 2.9e-3 apart).  The precompiled plans keep the shortcut (sampler/_plan.py): their
        # step sizes do not depend on the data.  Cost here: two elementwise operations per evaluation.
        noise = model_fn(x, tb)
        if not data:
            return noise
        x0 = (x - bc(ns.marginal_std(t), x) * noise) / bc(ns.marginal_alpha(t), x)
        if solver.correcting_x0_fn is not None:
            x0 = solver.correcting_x0_fn(x0, t)
        return x0
    return fn, data


def adaptive_sample(solver, x, order, t_T, t_0, atol=0.0078, rtol=0.05, solver_type="dpmsolver", h_init=0.05, theta=0.9,
                    t_err=1e-5):
    """x at t_T -> x at t_0; returns (x, nfe)."""
    if order not in (2, 3):
        raise ValueError("For adaptive step size solver, order must be 2 or 3, got {}".format(order))
    ns = solver.noise_schedule
    fn, data = _prediction_fn(solver)
    taylor = solver_type == "taylor"
    la, sg, lam = ns.marginal_log_mean_coeff, ns.marginal_std, ns.marginal_lambda

    def first(x, s, u, hu, m):                      # x at time u from (x, model_s): dpm_solver_first_update
        if data:
            return (sg(u) / sg(s)).to(x) * x - (torch.exp(la(u)) * torch.expm1(-hu)).to(x) * m
        return torch.exp(la(u) - la(s)).to(x) * x - (sg(u) * torch.expm1(hu)).to(x) * m

    def second(x, s, t, r1, m_s=None):              # singlestep second update; returns x_t, model_s, model_s1
        h = lam(t) - lam(s)
        s1 = ns.inverse_lambda(lam(s) + r1 * h)
        m_s = fn(x, s) if m_s is None else m_s
        m_s1 = fn(first(x, s, s1, r1 * h, m_s), s1)
        a_t, s_t = torch.exp(la(t)), sg(t)
        phi_1 = torch.expm1(-h) if data else torch.expm1(h)
        if data:
            c = -(0.5 / r1) * (a_t * phi_1) if not taylor else (1.0 / r1) * (a_t * (phi_1 / h + 1.0))
        else:
            c = -(0.5 / r1) * (s_t * phi_1) if not taylor else -(1.0 / r1) * (s_t * (phi_1 / h - 1.0))
        return first(x, s, t, h, m_s) + c.to(x) * (m_s1 - m_s), m_s, m_s1

    def third(x, s, t, r1, r2, m_s, m_s1):          # singlestep third update with model_s / model_s1 given
        h = lam(t) - lam(s)
        s2 = ns.inverse_lambda(lam(s) + r2 * h)
        a_t, s_t = torch.exp(la(t)), sg(t)
        if data:
            phi_1 = torch.expm1(-h)
            phi_22 = torch.expm1(-r2 * h) / (r2 * h) + 1.0
            phi_2 = phi_1 / h + 1.0
            x_s2 = first(x, s, s2, r2 * h, m_s) + (r2 / r1 * (torch.exp(la(s2)) * phi_22)).to(x) * (m_s1 - m_s)
        else:
            phi_1 = torch.expm1(h)
            phi_22 = torch.expm1(r2 * h) / (r2 * h) - 1.0
            phi_2 = phi_1 / h - 1.0
            x_s2 = first(x, s, s2, r2 * h, m_s) - (r2 / r1 * (sg(s2) * phi_22)).to(x) * (m_s1 - m_s)
        phi_3 = phi_2 / h - 0.5
        m_s2 = fn(x_s2, s2)
        amp = a_t if data else -s_t
        if not taylor:
            return first(x, s, t, h, m_s) + ((1.0 / r2) * (amp * phi_2)).to(x) * (m_s2 - m_s)
        D1_0 = (1.0 / r1) * (m_s1 - m_s)
        D1_1 = (1.0 / r2) * (m_s2 - m_s)
        D1 = (r2 * D1_0 - r1 * D1_1) / (r2 - r1)
        D2 = 2.0 * (D1_1 - D1_0) / (r2 - r1)
        return first(x, s, t, h, m_s) + (amp * phi_2).to(x) * D1 - ((a_t if data else s_t) * phi_3).to(x) * D2

    s = t_T * torch.ones((1,), dtype=torch.float32, device=x.device)
    lambda_s = lam(s)
    lambda_0 = lam(t_0 * torch.ones_like(s))
    h = h_init * torch.ones_like(s)
    x_prev = x
    nfe = 0
    r1, r2 = (0.5, None) if order == 2 else (1.0 / 3.0, 2.0 / 3.0)
    while float(torch.abs(s - t_0).mean()) > t_err:
        t = ns.inverse_lambda(lambda_s + h)
        if order == 2:
            m_s = fn(x, s)
            x_lower = first(x, s, t, lam(t) - lam(s), m_s)
            x_higher, _, _ = second(x, s, t, r1, m_s)
        else:
            x_lower, m_s, m_s1 = second(x, s, t, r1)
            x_higher = third(x, s, t, r1, r2, m_s, m_s1)
        delta = torch.max(torch.ones_like(x) * atol, rtol * torch.max(torch.abs(x_lower), torch.abs(x_prev)))
        v = (x_higher - x_lower) / delta
        E = torch.sqrt(torch.square(v.reshape((v.shape[0], -1))).mean(dim=-1, keepdim=True)).max()
        if bool(E <= 1.0):
            x, s, x_prev = x_higher, t, x_lower
            lambda_s = lam(s)
        h = torch.min(theta * h * torch.float_power(E, -1.0 / order).float(), lambda_0 - lambda_s)
        nfe += order
    return x, nfe
