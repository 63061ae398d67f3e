"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's
sampler loops `sampler/dpm_solver.py` (DPM-Solver++ multistep) and `sampler/uni_pc.py`
(UniPC B(h) multistep) together with their noise schedule and model wrapper.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
file.  It follows the reference op-for-op in float32 torch arithmetic (same order of
operations, same x0 -> noise -> x0 round trip through the model wrapper), so its
outputs agree with the imported reference to float32 rounding.  Pinned by
tools/make_golden.py -> tests/golden/sampler_*.npz (captured from the imported
reference; the reference has no tests of its own: SURVEY.md §4).

File:line citations are relative to the reference checkout.
"""
import torch


def piecewise_linear(x, xp, yp):
    """interpolate_fn, sampler/dpm_solver.py:1253-1292 (dup uni_pc.py:679-718): piecewise
    linear through (xp, yp) (xp ascending), linear extrapolation with the outermost
    segment beyond either end.  x: [N]; xp, yp: [K]."""
    K = xp.shape[0]
    i = torch.searchsorted(xp, x.contiguous(), right=False) - 1
    i = i.clamp(0, K - 2)
    x0, x1, y0, y1 = xp[i], xp[i + 1], yp[i], yp[i + 1]
    return y0 + (x - x0) * (y1 - y0) / (x1 - x0)


class Schedule:
    """NoiseScheduleVP('discrete', betas=...), sampler/dpm_solver.py:6-167 and
    sampler/uni_pc.py:6-152.  `clip` reproduces dpm_solver's numerical_clip_alpha
    (:114-125, lambda clipped at -5.1); uni_pc has no clip."""

    def __init__(self, betas, clip=False, dtype=torch.float32):
        log_alphas = 0.5 * torch.log(1 - betas).cumsum(dim=0)
        if clip:
            log_sigmas = 0.5 * torch.log(1.0 - torch.exp(2.0 * log_alphas))
            lambs = log_alphas - log_sigmas
            idx = int(torch.searchsorted(torch.flip(lambs, [0]), torch.tensor(-5.1, dtype=lambs.dtype)))
            if idx > 0:
                log_alphas = log_alphas[:-idx]
        self.T = 1.0
        self.log_alpha_array = log_alphas.to(dtype)
        self.total_N = self.log_alpha_array.shape[0]
        self.t_array = torch.linspace(0.0, 1.0, self.total_N + 1)[1:].to(dtype)

    def log_alpha(self, t):   # marginal_log_mean_coeff
        return piecewise_linear(t.reshape(-1), self.t_array, self.log_alpha_array)

    def alpha(self, t):       # marginal_alpha
        return torch.exp(self.log_alpha(t))

    def sigma(self, t):       # marginal_std
        return torch.sqrt(1.0 - torch.exp(2.0 * self.log_alpha(t)))

    def lam(self, t):         # marginal_lambda
        la = self.log_alpha(t)
        return la - 0.5 * torch.log(1.0 - torch.exp(2.0 * la))

    def inverse_lambda(self, lamb):
        log_alpha = -0.5 * torch.logaddexp(torch.zeros((1,)), -2.0 * lamb)
        return piecewise_linear(log_alpha.reshape(-1), torch.flip(self.log_alpha_array, [0]),
                                torch.flip(self.t_array, [0]))


class ContinuousSchedule(Schedule):
    """NoiseScheduleVP('linear' | 'cosine'): the continuous-time VP schedules as the reference evaluates them - closed
    forms in float32 torch arithmetic (dpm_solver.py:108-111,133-134,160-163; uni_pc.py:67-100,125-149).  total_N = 1000,
    T = 1 (0.9946 for 'cosine'); the network is called with t itself (dpm_solver.py:271-280)."""

    def __init__(self, kind, beta_0=0.1, beta_1=20.0):
        import math
        assert kind in ("linear", "cosine")
        self.kind, self.beta_0, self.beta_1 = kind, beta_0, beta_1
        self.total_N = 1000
        self.cosine_s = 0.008
        self.cosine_log_alpha_0 = math.log(math.cos(self.cosine_s / (1.0 + self.cosine_s) * math.pi / 2.0))
        self.T = 0.9946 if kind == "cosine" else 1.0

    def log_alpha(self, t):
        import math
        t = t.reshape(-1)
        if self.kind == "linear":
            return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.log(torch.cos((t + self.cosine_s) / (1.0 + self.cosine_s) * math.pi / 2.0)) - self.cosine_log_alpha_0

    def inverse_lambda(self, lamb):
        import math
        if self.kind == "linear":
            tmp = 2.0 * (self.beta_1 - self.beta_0) * torch.logaddexp(-2.0 * lamb, torch.zeros((1,)).to(lamb))
            delta = self.beta_0 ** 2 + tmp
            return tmp / (torch.sqrt(delta) + self.beta_0) / (self.beta_1 - self.beta_0)
        log_alpha = -0.5 * torch.logaddexp(-2.0 * lamb, torch.zeros((1,)).to(lamb))
        return torch.arccos(torch.exp(log_alpha + self.cosine_log_alpha_0)) * 2.0 * (1.0 + self.cosine_s) / math.pi - self.cosine_s


def _schedule(betas, clip, schedule):
    """`schedule`: None (discrete, from betas) or ('linear' | 'cosine', beta_0, beta_1)."""
    if schedule is None or schedule[0] == "discrete":
        return Schedule(betas, clip=clip)
    return ContinuousSchedule(*schedule)


def _bcast(v, x):
    return v.reshape((-1,) + (1,) * (x.dim() - 1))


def dynamic_thresholding(x0, ratio=0.995, max_val=1.0):
    """dynamic_thresholding_fn, dpm_solver.py:416-425 / uni_pc.py:268-277."""
    s = torch.quantile(torch.abs(x0).reshape((x0.shape[0], -1)), ratio, dim=1)
    s = torch.maximum(s, max_val * torch.ones_like(s)).reshape((-1,) + (1,) * (x0.dim() - 1))
    return torch.clamp(x0, -s, s) / s


def wrap_x_start_model(model, ns, x0_fn=None):
    """model_wrapper(model, ns, model_type='x_start') followed by the solver's
    data_prediction_fn: t_input = (t - 1/N) * N (dpm_solver.py:271-280); noise =
    (x - alpha_t * x0) / sigma_t (:290-292); x0 = (x - sigma_t * noise) / alpha_t (:433-442).
    The broadcast over batch is the dpm_solver form (expand_dims); uni_pc.py:189-191 omits
    it and only runs at B=1 where both forms coincide."""
    def data_prediction(x, t):
        tb = t.expand(x.shape[0])
        t_input = tb if isinstance(ns, ContinuousSchedule) else (tb - 1.0 / ns.total_N) * ns.total_N
        out = model(x, t_input)
        a, s = ns.alpha(tb), ns.sigma(tb)
        noise = (x - _bcast(a, x) * out) / _bcast(s, x)
        a1, s1 = ns.alpha(t), ns.sigma(t)
        x0 = (x - s1 * noise) / a1
        return x0 if x0_fn is None else x0_fn(x0, t)      # correcting_x0_fn (dpm_solver.py:443-444; uni_pc.py:292-293 passes x0 only)
    return data_prediction


def guided_noise_fn(model, ns, guidance_type="uncond", condition=None, unconditional_condition=None, guidance_scale=1.0,
                    classifier_fn=None, model_type="x_start"):
    """model_wrapper(model, ns, model_type='x_start', guidance_type='classifier' | 'classifier-free', ...) ->
    model_fn(x, t): dpm_solver.py:282-330.  `model(x, t_input[, cond])` predicts x0."""
    def noise_pred(x, t, cond=None):      # dpm_solver.py:282-298, every model_type
        t_input = t if isinstance(ns, ContinuousSchedule) else (t - 1.0 / ns.total_N) * ns.total_N
        out = model(x, t_input) if cond is None else model(x, t_input, cond)
        if model_type == "noise":
            return out
        if model_type == "x_start":
            return (x - _bcast(ns.alpha(t), x) * out) / _bcast(ns.sigma(t), x)
        if model_type == "v":
            return _bcast(ns.alpha(t), x) * out + _bcast(ns.sigma(t), x) * x
        assert model_type == "score"
        return -_bcast(ns.sigma(t), x) * out

    def model_fn(x, t):
        if guidance_type == "uncond":
            return noise_pred(x, t)
        if guidance_type == "classifier":
            t_input = t if isinstance(ns, ContinuousSchedule) else (t - 1.0 / ns.total_N) * ns.total_N
            with torch.enable_grad():
                x_in = x.detach().requires_grad_(True)
                grad = torch.autograd.grad(classifier_fn(x_in, t_input, condition).sum(), x_in)[0]
            return noise_pred(x, t) - guidance_scale * _bcast(ns.sigma(t), x) * grad
        if guidance_scale == 1.0 or unconditional_condition is None:
            return noise_pred(x, t, cond=condition)
        nu, n = noise_pred(torch.cat([x] * 2), torch.cat([t] * 2), cond=torch.cat([unconditional_condition, condition])).chunk(2)
        return nu + guidance_scale * (n - nu)
    return model_fn


def time_steps(ns, skip_type, t_T, t_0, N):
    """get_time_steps, dpm_solver.py:453-480."""
    if skip_type == "time_uniform":
        return torch.linspace(t_T, t_0, N + 1)
    if skip_type == "time_quadratic":
        return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2)
    if skip_type == "logSNR":
        lT = ns.lam(torch.tensor(t_T))
        l0 = ns.lam(torch.tensor(t_0))
        return ns.inverse_lambda(torch.linspace(lT.item(), l0.item(), N + 1))
    raise ValueError("Unsupported skip_type %r" % (skip_type,))


# ----------------------------------------------------------------------------- DPM-Solver++
def _dpmpp_update(ns, x, m_list, t_list, t, order, taylor=False):
    """multistep_dpm_solver_update for algorithm_type='dpmsolver++', solver_type='dpmsolver':
    first (:547-580), second (:796-831), third (:854-889) order."""
    t0 = t_list[-1]
    lam0, lam_t = ns.lam(t0), ns.lam(t)
    sig0, sig_t = ns.sigma(t0), ns.sigma(t)
    alpha_t = torch.exp(ns.log_alpha(t))
    h = lam_t - lam0
    phi_1 = torch.expm1(-h)
    if order == 1:
        return sig_t / sig0 * x - alpha_t * phi_1 * m_list[-1]
    if order == 2:
        m1, m0 = m_list[-2], m_list[-1]
        h_0 = lam0 - ns.lam(t_list[-2])
        r0 = h_0 / h
        D1_0 = (1.0 / r0) * (m0 - m1)
        if taylor:                 # solver_type='taylor', dpm_solver.py:825-829
            return (sig_t / sig0) * x - (alpha_t * phi_1) * m0 + (alpha_t * (phi_1 / h + 1.0)) * D1_0
        return (sig_t / sig0) * x - (alpha_t * phi_1) * m0 - 0.5 * (alpha_t * phi_1) * D1_0
    if order == 3:
        m2, m1, m0 = m_list
        lam1, lam2 = ns.lam(t_list[-2]), ns.lam(t_list[-3])
        h_1, h_0 = lam1 - lam2, lam0 - lam1
        r0, r1 = h_0 / h, h_1 / h
        D1_0 = (1.0 / r0) * (m0 - m1)
        D1_1 = (1.0 / r1) * (m1 - m2)
        D1 = D1_0 + (r0 / (r0 + r1)) * (D1_0 - D1_1)
        D2 = (1.0 / (r0 + r1)) * (D1_0 - D1_1)
        phi_2 = phi_1 / h + 1.0
        phi_3 = phi_2 / h - 0.5
        return (sig_t / sig0) * x - (alpha_t * phi_1) * m0 + (alpha_t * phi_2) * D1 - (alpha_t * phi_3) * D2
    raise ValueError("Solver order must be 1 or 2 or 3, got %r" % (order,))


def _dpm_noise_update(ns, x, m_list, t_list, t, order, taylor=False):
    """multistep_dpm_solver_update for algorithm_type='dpmsolver' (updates on the noise prediction), solver_type='dpmsolver':
    first (dpm_solver.py:581-592), second (:841-847), third (:895-904) order."""
    t0 = t_list[-1]
    lam0, lam_t = ns.lam(t0), ns.lam(t)
    la0, la_t = ns.log_alpha(t0), ns.log_alpha(t)
    sig_t = ns.sigma(t)
    h = lam_t - lam0
    phi_1 = torch.expm1(h)
    if order == 1:
        return torch.exp(la_t - la0) * x - (sig_t * phi_1) * m_list[-1]
    if order == 2:
        m1, m0 = m_list[-2], m_list[-1]
        r0 = (lam0 - ns.lam(t_list[-2])) / h
        D1_0 = (1.0 / r0) * (m0 - m1)
        if taylor:                 # solver_type='taylor', dpm_solver.py:848-851
            return torch.exp(la_t - la0) * x - (sig_t * phi_1) * m0 - (sig_t * (phi_1 / h - 1.0)) * D1_0
        return torch.exp(la_t - la0) * x - (sig_t * phi_1) * m0 - 0.5 * (sig_t * phi_1) * D1_0
    if order == 3:
        m2, m1, m0 = m_list
        lam1, lam2 = ns.lam(t_list[-2]), ns.lam(t_list[-3])
        h_1, h_0 = lam1 - lam2, lam0 - lam1
        r0, r1 = h_0 / h, h_1 / h
        D1_0 = (1.0 / r0) * (m0 - m1)
        D1_1 = (1.0 / r1) * (m1 - m2)
        D1 = D1_0 + (r0 / (r0 + r1)) * (D1_0 - D1_1)
        D2 = (1.0 / (r0 + r1)) * (D1_0 - D1_1)
        phi_2 = phi_1 / h - 1.0
        phi_3 = phi_2 / h - 0.5
        return torch.exp(la_t - la0) * x - (sig_t * phi_1) * m0 - (sig_t * phi_2) * D1 - (sig_t * phi_3) * D2
    raise ValueError("Solver order must be 1 or 2 or 3, got %r" % (order,))


def _singlestep_update(ns, fn, x, s, t, order, r1, r2, noise, taylor):
    """singlestep_dpm_solver_update: first (dpm_solver.py:547-592), second (:594-676), third (:678-794) order from time s to
    time t; fn = model_fn (data prediction for 'dpmsolver++', noise prediction for 'dpmsolver')."""
    lam_s, lam_t = ns.lam(s), ns.lam(t)
    h = lam_t - lam_s
    la = ns.log_alpha
    sg = ns.sigma

    def first(u, hu, m):      # x at time u from (x, model_s)
        if noise:
            return torch.exp(la(u) - la(s)) * x - (sg(u) * torch.expm1(hu)) * m
        return (sg(u) / sg(s)) * x - (torch.exp(la(u)) * torch.expm1(-hu)) * m
    m_s = fn(x, s)
    if order == 1:
        return first(t, h, m_s)
    s1 = ns.inverse_lambda(lam_s + r1 * h)
    m_s1 = fn(first(s1, r1 * h, m_s), s1)
    alpha_t, sig_t = torch.exp(la(t)), sg(t)
    phi_1 = torch.expm1(h) if noise else torch.expm1(-h)
    if order == 2:
        if noise:
            corr = -(0.5 / r1) * (sig_t * phi_1) * (m_s1 - m_s) if not taylor else -(1.0 / r1) * (sig_t * (phi_1 / h - 1.0)) * (m_s1 - m_s)
        else:
            corr = -(0.5 / r1) * (alpha_t * phi_1) * (m_s1 - m_s) if not taylor else (1.0 / r1) * (alpha_t * (phi_1 / h + 1.0)) * (m_s1 - m_s)
        return first(t, h, m_s) + corr
    s2 = ns.inverse_lambda(lam_s + r2 * h)
    if noise:
        phi_22 = torch.expm1(r2 * h) / (r2 * h) - 1.0
        phi_2 = phi_1 / h - 1.0
        x_s2 = first(s2, r2 * h, m_s) - r2 / r1 * (sg(s2) * phi_22) * (m_s1 - m_s)
    else:
        phi_22 = torch.expm1(-r2 * h) / (r2 * h) + 1.0
        phi_2 = phi_1 / h + 1.0
        x_s2 = first(s2, r2 * h, m_s) + r2 / r1 * (torch.exp(la(s2)) * phi_22) * (m_s1 - m_s)
    phi_3 = phi_2 / h - 0.5
    m_s2 = fn(x_s2, s2)
    if not taylor:
        if noise:
            return first(t, h, m_s) - (1.0 / r2) * (sig_t * phi_2) * (m_s2 - m_s)
        return first(t, h, m_s) + (1.0 / r2) * (alpha_t * phi_2) * (m_s2 - m_s)
    D1_0 = (1.0 / r1) * (m_s1 - m_s)
    D1_1 = (1.0 / r2) * (m_s2 - m_s)
    D1 = (r2 * D1_0 - r1 * D1_1) / (r2 - r1)
    D2 = 2.0 * (D1_1 - D1_0) / (r2 - r1)
    if noise:
        return first(t, h, m_s) - (sig_t * phi_2) * D1 - (sig_t * phi_3) * D2
    return first(t, h, m_s) + (alpha_t * phi_2) * D1 - (alpha_t * phi_3) * D2


def singlestep_orders(ns, steps, order, skip_type, t_T, t_0, fixed=False):
    """get_orders_and_timesteps_for_singlestep_solver (dpm_solver.py:482-539) / the 'singlestep_fixed' grid (:1217-1220)."""
    if fixed:
        K = steps // order
        return time_steps(ns, skip_type, t_T, t_0, K), [order] * K
    if order == 3:
        K = steps // 3 + 1
        orders = [3] * (K - 2) + [2, 1] if steps % 3 == 0 else ([3] * (K - 1) + [1] if steps % 3 == 1 else [3] * (K - 1) + [2])
    elif order == 2:
        K = steps // 2 if steps % 2 == 0 else steps // 2 + 1
        orders = [2] * K if steps % 2 == 0 else [2] * (K - 1) + [1]
    else:
        K, orders = 1, [1] * steps
    if skip_type == "logSNR":
        return time_steps(ns, skip_type, t_T, t_0, K), orders
    return time_steps(ns, skip_type, t_T, t_0, steps)[torch.cumsum(torch.tensor([0] + orders), 0)], orders


def _adaptive(ns, fn, x, order, t_T, t_0, noise, taylor, h_init=0.05, atol=0.0078, rtol=0.05, theta=0.9, t_err=1e-5):
    """dpm_solver_adaptive, dpm_solver.py:906-1010: the lower-order update (order - 1) estimates the error of the order-th
    singlestep update; a step is accepted when the scaled error is <= 1, the next step size follows theta h E^(-1/order).
    (The reference hands model_s / model_s1 from the lower to the higher update; re-evaluating them gives the same bits.)
    Returns (x, nfe)."""
    assert order in (2, 3)
    s = t_T * torch.ones((1,))
    lam_s = ns.lam(s)
    lam_0 = ns.lam(t_0 * torch.ones_like(s))
    h = h_init * torch.ones_like(s)
    x_prev, nfe = x, 0
    r1, r2 = (0.5, None) if order == 2 else (1.0 / 3.0, 2.0 / 3.0)
    while torch.abs(s - t_0).mean() > t_err:
        t = ns.inverse_lambda(lam_s + h)
        x_lower = _singlestep_update(ns, fn, x, s, t, order - 1, r1, None, noise, taylor)
        x_higher = _singlestep_update(ns, fn, x, s, t, order, r1, r2, noise, taylor)
        delta = torch.max(torch.ones_like(x) * atol, rtol * torch.max(torch.abs(x_lower), torch.abs(x_prev)))
        v = (x_higher - x_lower) / delta
        E = torch.sqrt(torch.square(v.reshape((v.shape[0], -1))).mean(dim=-1, keepdim=True)).max()
        if torch.all(E <= 1.0):
            x, s, x_prev = x_higher, t, x_lower
            lam_s = ns.lam(s)
        h = torch.min(theta * h * torch.float_power(E, -1.0 / order).float(), lam_0 - lam_s)
        nfe += order
    return x, nfe


def wrap_x_start_noise(model, ns):
    """model_wrapper(model, ns, model_type='x_start') alone: the NOISE prediction the algorithm_type='dpmsolver' updates
    consume - noise = (x - alpha_t * x0) / sigma_t (dpm_solver.py:290-292)."""
    def noise_prediction(x, t):
        tb = t.expand(x.shape[0])
        t_input = tb if isinstance(ns, ContinuousSchedule) else (tb - 1.0 / ns.total_N) * ns.total_N
        out = model(x, t_input)
        a, s = ns.alpha(tb), ns.sigma(tb)
        return (x - _bcast(a, x) * out) / _bcast(s, x)
    return noise_prediction


def dpm_solver_pp_sample(model, betas, x, steps=20, order=2, skip_type="time_uniform",
                         lower_order_final=True, return_intermediate=False, t_start=None, t_end=None,
                         denoise_to_zero=False, schedule=None, algorithm_type="dpmsolver++", x0_fn=None, xt_fn=None,
                         guidance=None, solver_type="dpmsolver", method="multistep", atol=0.0078, rtol=0.05):
    """DPM_Solver(model_fn, ns, algorithm_type).sample(x, steps, order, skip_type,
    method='multistep'), dpm_solver.py:1047-1245 (multistep branch :1171-1213).
    `model(x, t_input)` is the raw x0-prediction network.  algorithm_type='dpmsolver': the same loop on the noise
    prediction (model_fn = noise_prediction_fn, dpm_solver.py:390-392)."""
    ns = _schedule(betas, True, schedule)
    data_fn = wrap_x_start_model(model, ns, x0_fn)
    if guidance is not None:       # a guided wrapper (dict of guided_noise_fn's keywords): model_fn is its noise prediction
        g_noise = guided_noise_fn(model, ns, **guidance)

        def data_fn(x, t):         # data_prediction_fn, dpm_solver.py:433-445
            tb = t.expand(x.shape[0])
            x0 = (x - ns.sigma(t) * g_noise(x, tb)) / ns.alpha(t)
            return x0 if x0_fn is None else x0_fn(x0, t)
    fix = (lambda x, t, step: x) if xt_fn is None else xt_fn       # correcting_xt_fn (:1180-1181, 1188-1189, 1203-1204, 1237-1238)
    if algorithm_type == "dpmsolver++":
        fn = data_fn
    elif guidance is not None:                       # model_fn = noise_prediction_fn (dpm_solver.py:427-431) of the guided / typed wrapper
        fn = lambda x, t: g_noise(x, t.expand(x.shape[0]))
    else:
        fn = wrap_x_start_noise(model, ns)
    _upd = globals()["_dpmpp_update"] if algorithm_type == "dpmsolver++" else _dpm_noise_update
    _dpmpp_update = lambda *a: _upd(*a, taylor=solver_type == "taylor")
    # dpm_solver.py:1157-1158: t_0 = 1/N unless t_end is given, t_T = T unless t_start is given
    t_0 = 1.0 / ns.total_N if t_end is None else t_end
    t_T = ns.T if t_start is None else t_start
    assert steps >= order
    if method == "adaptive":       # dpm_solver.py:1162-1170
        assert not return_intermediate, "Cannot use adaptive solver when saving intermediate values"
        x, _ = _adaptive(ns, fn, x, order, t_T, t_0, algorithm_type != "dpmsolver++", solver_type == "taylor", atol=atol, rtol=rtol)
        return data_fn(x, torch.ones((1,)) * t_0) if denoise_to_zero else x
    if method != "multistep":      # 'singlestep' / 'singlestep_fixed', dpm_solver.py:1214-1232 (no start point in the intermediates)
        outer, orders = singlestep_orders(ns, steps, order, skip_type, t_T, t_0, method == "singlestep_fixed")
        inter = []
        for step, k in enumerate(orders):
            s_, t = outer[step], outer[step + 1]
            lam_in = ns.lam(time_steps(ns, skip_type, s_.item(), t.item(), k))
            h_in = lam_in[-1] - lam_in[0]
            r1 = None if k <= 1 else (lam_in[1] - lam_in[0]) / h_in
            r2 = None if k <= 2 else (lam_in[2] - lam_in[0]) / h_in
            x = fix(_singlestep_update(ns, fn, x, s_, t, k, r1, r2, algorithm_type != "dpmsolver++", solver_type == "taylor"), t, step)
            inter.append(x)
        if denoise_to_zero:
            x = fix(data_fn(x, torch.ones((1,)) * t_0), torch.ones((1,)) * t_0, len(orders))
            inter.append(x)
        return (x, inter) if return_intermediate else x
    ts = time_steps(ns, skip_type, t_T, t_0, steps)
    inter = []
    t = ts[0]
    t_list, m_list = [t], [fn(x, t)]
    x = fix(x, t, 0)
    for step in range(1, order):
        t = ts[step]
        x = fix(_dpmpp_update(ns, x, m_list, t_list, t, step), t, step)
        inter.append(x)
        t_list.append(t)
        m_list.append(fn(x, t))
    for step in range(order, steps + 1):
        t = ts[step]
        step_order = min(order, steps + 1 - step) if (lower_order_final and steps < 10) else order
        x = fix(_dpmpp_update(ns, x, m_list, t_list, t, step_order), t, step)
        inter.append(x)
        t_list = t_list[1:] + [t]
        m_list = m_list[1:] + [None]
        if step < steps:
            m_list[-1] = fn(x, t)
    if denoise_to_zero:            # dpm_solver.py:1234-1240, :541-545: x0 prediction at t_0 (one more evaluation)
        x = fix(data_fn(x, torch.ones((1,)) * t_0), torch.ones((1,)) * t_0, steps + 1)
        inter.append(x)
    return (x, inter) if return_intermediate else x


# ----------------------------------------------------------------------------- UniPC
def _unipc_bh_update(ns, fn, x, m_list, t_list, t, order, variant, use_corrector, predict_x0=True):
    """multistep_uni_pc_bh_update, uni_pc.py:471-588 (predict_x0=False: the noise form, :503, :569-587)."""
    t = t.reshape(-1)
    t0 = t_list[-1]
    lam0, lam_t = ns.lam(t0), ns.lam(t)
    m0 = m_list[-1]
    sig0, sig_t = ns.sigma(t0), ns.sigma(t)
    alpha_t = torch.exp(ns.log_alpha(t))
    h = lam_t - lam0
    rks, D1s = [], []
    for i in range(1, order):
        rk = (ns.lam(t_list[-(i + 1)]) - lam0) / h
        rks.append(rk)
        D1s.append((m_list[-(i + 1)] - m0) / rk)
    rks.append(1.0)
    rks = torch.tensor([float(r) for r in rks])
    hh = -h if predict_x0 else h
    h_phi_1 = torch.expm1(hh)
    h_phi_k = h_phi_1 / hh - 1
    B_h = hh if variant == "bh1" else torch.expm1(hh)
    R, b = [], []
    fact = 1
    for i in range(1, order + 1):
        R.append(torch.pow(rks, i - 1))
        b.append(h_phi_k * fact / B_h)
        fact *= (i + 1)
        h_phi_k = h_phi_k / hh - 1 / fact
    R = torch.stack(R)
    b = torch.cat(b)
    rhos_p = None
    if D1s:
        D1s = torch.stack(D1s, dim=1)
        rhos_p = torch.tensor([0.5]) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
    else:
        D1s = None
    if use_corrector:
        rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
    amp = alpha_t if predict_x0 else sig_t
    if predict_x0:
        x_t_ = sig_t / sig0 * x - alpha_t * h_phi_1 * m0
    else:
        x_t_ = torch.exp(ns.log_alpha(t) - ns.log_alpha(t0)) * x - sig_t * h_phi_1 * m0
    pred = torch.einsum("k,bkct->bct", rhos_p, D1s) if D1s is not None else 0
    x_t = x_t_ - amp * B_h * pred
    m_t = None
    if use_corrector:
        m_t = fn(x_t, t)
        corr = torch.einsum("k,bkct->bct", rhos_c[:-1], D1s) if D1s is not None else 0
        x_t = x_t_ - amp * B_h * (corr + rhos_c[-1] * (m_t - m0))
    return x_t, m_t


def _unipc_vary_update(ns, fn, x, m_list, t_list, t, order, use_corrector, predict_x0=True):
    """multistep_uni_pc_vary_update, uni_pc.py:368-469 (variant='vary_coeff'; predict_x0=False: :417, :448-468): residual weights
    from the inverse of C[i, k] = rks[i]^k / (k+1)! instead of the B(h) systems.  The corrector's last term indexes
    A_c with the loop variable `k` left over from the residual loop (:444-447), reproduced as written."""
    t = t.reshape(-1)
    t0 = t_list[-1]
    lam0, lam_t = ns.lam(t0), ns.lam(t)
    m0 = m_list[-1]
    sig0, sig_t = ns.sigma(t0), ns.sigma(t)
    alpha_t = torch.exp(ns.log_alpha(t))
    h = lam_t - lam0
    rks, D1s = [], []
    for i in range(1, order):
        rk = (ns.lam(t_list[-(i + 1)]) - lam0) / h
        rks.append(rk)
        D1s.append((m_list[-(i + 1)] - m0) / rk)
    rks.append(1.0)
    rks = torch.tensor([float(r) for r in rks])
    K = len(rks)
    cols, col = [], torch.ones_like(rks)
    for k in range(1, K + 1):
        cols.append(col)
        col = col * rks / (k + 1)
    Cm = torch.stack(cols, dim=1)
    A_p = torch.linalg.inv(Cm[:-1, :-1]) if D1s else None
    if D1s:
        D1s = torch.stack(D1s, dim=1)
    A_c = torch.linalg.inv(Cm) if use_corrector else None
    hh = -h if predict_x0 else h
    h_phi_1 = torch.expm1(hh)
    h_phi_ks, fact, h_phi_k = [], 1, h_phi_1
    for k in range(1, K + 2):
        h_phi_ks.append(h_phi_k)
        h_phi_k = h_phi_k / hh - 1 / fact
        fact *= (k + 1)
    amp = alpha_t if predict_x0 else sig_t
    if predict_x0:
        x_t_ = sig_t / sig0 * x - alpha_t * h_phi_1 * m0
    else:
        x_t_ = torch.exp(ns.log_alpha(t) - ns.log_alpha(t0)) * x - sig_t * h_phi_1 * m0
    x_t = x_t_
    if len(D1s) > 0:
        for k in range(K - 1):
            x_t = x_t - amp * h_phi_ks[k + 1] * torch.einsum("bkct,k->bct", D1s, A_p[k])
    m_t = None
    if use_corrector:
        m_t = fn(x_t, t)
        D1_t = m_t - m0
        x_t = x_t_
        k = 0
        for k in range(K - 1):
            x_t = x_t - amp * h_phi_ks[k + 1] * torch.einsum("bkct,k->bct", D1s, A_c[k][:-1])
        x_t = x_t - amp * h_phi_ks[K] * (D1_t * A_c[k][-1])
    return x_t, m_t


def _unipc_update(ns, fn, x, m_list, t_list, t, order, variant, use_corrector, predict_x0=True):
    """multistep_uni_pc_update dispatch, uni_pc.py:357-366."""
    if "bh" in variant:
        return _unipc_bh_update(ns, fn, x, m_list, t_list, t, order, variant, use_corrector, predict_x0)
    assert variant == "vary_coeff"
    return _unipc_vary_update(ns, fn, x, m_list, t_list, t, order, use_corrector, predict_x0)


def unipc_sample(model, betas, x, steps=20, order=2, skip_type="time_uniform", variant="bh2",
                 lower_order_final=True, return_intermediate=False, t_start=None, t_end=None, denoise_to_zero=False,
                 schedule=None, x0_fn=None, xt_fn=None, algorithm_type="data_prediction"):
    """UniPC(model_fn, ns, algorithm_type, variant=...).sample(x, steps, order, skip_type, 'multistep'),
    uni_pc.py:590-672.  algorithm_type='noise_prediction': the same loop on the noise prediction (:266, :296-303)."""
    ns = _schedule(betas, False, schedule)
    data_fn = wrap_x_start_model(model, ns, None if x0_fn is None else (lambda x0, t: x0_fn(x0)))
    px0 = algorithm_type == "data_prediction"
    fn = data_fn if px0 else wrap_x_start_noise(model, ns)
    _unipc_update = lambda *a: globals()["_unipc_update"](*a, predict_x0=px0)
    fix = (lambda x, t, step: x) if xt_fn is None else xt_fn       # correcting_xt_fn (uni_pc.py:615-616, 626-627, 646-647, 664-665)
    t_0 = 1.0 / ns.total_N if t_end is None else t_end        # uni_pc.py:596-597
    t_T = ns.T if t_start is None else t_start
    assert steps >= order
    ts = time_steps(ns, skip_type, t_T, t_0, steps)
    inter = []
    t = ts[0]
    t_list, m_list = [t], [fn(x, t)]
    x = fix(x, t, 0)
    for step in range(1, order):
        t = ts[step]
        x, m_x = _unipc_update(ns, fn, x, m_list, t_list, t, step, variant, True)
        x = fix(x, t, step)
        inter.append(x)
        t_list.append(t)
        m_list.append(m_x)
    for step in range(order, steps + 1):
        t = ts[step]
        step_order = min(order, steps + 1 - step) if lower_order_final else order
        x, m_x = _unipc_update(ns, fn, x, m_list, t_list, t, step_order, variant, step != steps)
        x = fix(x, t, step)
        inter.append(x)
        t_list = t_list[1:] + [t]
        m_list = m_list[1:] + [None]
        if step < steps:
            m_list[-1] = m_x
    if denoise_to_zero:            # uni_pc.py:660-666
        x = fix(data_fn(x, torch.ones((1,)) * t_0), torch.ones((1,)) * t_0, steps + 1)
        inter.append(x)
    return (x, inter) if return_intermediate else x


def standin_model(x, t_input):
    """Analytic stand-in network for sampler known-answer tests:
    x0 = tanh(x/2) * (1 + 1e-6 * tau)."""
    return torch.tanh(x / 2) * (1 + 1e-6 * t_input.reshape((-1,) + (1,) * (x.dim() - 1)))


def standin_x0_fix(x0, t=None):
    """A correcting_x0_fn for the known-answer cases (dpm_solver passes (x0, t), uni_pc passes x0 only)."""
    return 0.98 * x0 if t is None else 0.98 * x0 + 0.01 * t


def standin_xt_fix(x, t, step):
    """A correcting_xt_fn for the known-answer cases: depends on all three arguments."""
    return x * (1.0 - 1e-3 * (step % 3)) + 1e-3 * t


def standin_cond_model(x, t_input, cond=None):
    """The stand-in network with an optional condition (guidance known-answer cases)."""
    out = standin_model(x, t_input)
    return out if cond is None else out + 0.1 * torch.tanh(cond)


def standin_classifier(x, t_input, cond):
    """log p_t(cond | x) of the guidance known-answer cases (differentiable in x)."""
    return -0.005 * ((x - cond) ** 2).reshape(x.shape[0], -1).sum(1)

