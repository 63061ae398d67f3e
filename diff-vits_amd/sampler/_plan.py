"""Shared machinery of the two sampler mirrors: the compiled loop (`dv_sampler_plan`, host
fp64 tables from libdvits_hip.so), its execution with an arbitrary Python callable (torch
ops on whatever device the tensors live on) and its native execution (hipGraph replay of
UNet schedule + fused update kernels) when the model is this package's UNet.
"""
import ctypes as C
import os
import importlib

import numpy as np
import torch


def _lib():
    return importlib.import_module("diff_vits_amd._lib")


class NoiseScheduleBase:
    """`NoiseScheduleVP('discrete', betas=... | alphas_cumprod=...)` (reference
    sampler/dpm_solver.py:6-167, sampler/uni_pc.py:6-152).  Arrays are stored in float32 like
    the reference; the look-ups interpolate them in float64 on the host.
    `NoiseScheduleVP('linear', continuous_beta_0=, continuous_beta_1=)` / `('cosine')`: the continuous-time VP schedules
    (dpm_solver.py:108-111,133-134,160-163; uni_pc.py:67-100) as closed forms; total_N = 1000, T = 1 (0.9946 for 'cosine')."""

    clip_lambda = None   # dpm_solver clips log-SNR at -5.1 (numerical_clip_alpha :114-125); uni_pc does not
    schedules = ("discrete", "linear")   # (uni_pc adds 'cosine')
    _COS_S = 0.008

    def __init__(self, schedule="discrete", betas=None, alphas_cumprod=None, continuous_beta_0=0.1,
                 continuous_beta_1=20.0, dtype=torch.float32):
        if schedule not in self.schedules:
            raise ValueError("Unsupported noise schedule {}. The schedule needs to be {}".format(
                schedule, " or ".join("'%s'" % s for s in self.schedules)))
        self.schedule = schedule
        self.beta_0, self.beta_1 = float(continuous_beta_0), float(continuous_beta_1)
        if schedule != "discrete":
            self.total_N = 1000
            self.T = 0.9946 if schedule == "cosine" else 1.0
            self._betas = np.zeros(2, dtype=np.float32)          # (unused by the native plan of a continuous schedule)
            self._cos_la0 = float(np.log(np.cos(self._COS_S / (1.0 + self._COS_S) * np.pi / 2.0)))
            return
        if betas is not None:
            b = torch.as_tensor(betas).detach().to("cpu", torch.float32)
            log_alphas = 0.5 * torch.log(1 - b).cumsum(dim=0)
        else:
            assert alphas_cumprod is not None
            ac = torch.as_tensor(alphas_cumprod).detach().to("cpu", torch.float32)
            log_alphas = 0.5 * torch.log(ac)
            b = 1 - torch.cat([ac[:1], ac[1:] / ac[:-1]])
        self._betas = b.numpy().astype(np.float32)
        if self.clip_lambda is not None:
            log_sigmas = 0.5 * torch.log(1.0 - torch.exp(2.0 * log_alphas))
            lambs = log_alphas - log_sigmas
            idx = int(torch.searchsorted(torch.flip(lambs, [0]), torch.tensor(self.clip_lambda, dtype=lambs.dtype)))
            if idx > 0:
                log_alphas = log_alphas[:-idx]
        self.T = 1.0
        self.log_alpha_array = log_alphas.reshape((1, -1)).to(dtype=dtype)
        self.total_N = self.log_alpha_array.shape[1]
        self.t_array = torch.linspace(0.0, 1.0, self.total_N + 1)[1:].reshape((1, -1)).to(dtype=dtype)
        self._xp = self.t_array[0].double().numpy()
        self._yp = self.log_alpha_array[0].double().numpy()

    @staticmethod
    def _interp(x, xp, yp):
        i = np.clip(np.searchsorted(xp, x, side="left") - 1, 0, len(xp) - 2)
        return yp[i] + (x - xp[i]) * (yp[i + 1] - yp[i]) / (xp[i + 1] - xp[i])

    def _la(self, t):
        t = np.asarray(torch.as_tensor(t).detach().cpu().double().reshape(-1))
        if self.schedule == "linear":
            return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        if self.schedule == "cosine":
            return np.log(np.cos((t + self._COS_S) / (1.0 + self._COS_S) * np.pi / 2.0)) - self._cos_la0
        return self._interp(t, self._xp, self._yp)

    @staticmethod
    def _like(v, t):
        t = torch.as_tensor(t)
        dt = t.dtype if t.is_floating_point() else torch.float32
        return torch.as_tensor(v, dtype=dt, device=t.device)

    def marginal_log_mean_coeff(self, t):
        return self._like(self._la(t), t)

    def marginal_alpha(self, t):
        return self._like(np.exp(self._la(t)), t)

    def marginal_std(self, t):
        return self._like(np.sqrt(1.0 - np.exp(2.0 * self._la(t))), t)

    def marginal_lambda(self, t):
        la = self._la(t)
        return self._like(la - 0.5 * np.log(1.0 - np.exp(2.0 * la)), t)

    def inverse_lambda(self, lamb):
        lam = np.asarray(torch.as_tensor(lamb).detach().cpu().double().reshape(-1))
        lse = np.logaddexp(0.0, -2.0 * lam)
        if self.schedule == "linear":        # dpm_solver.py:160-163
            tmp = 2.0 * (self.beta_1 - self.beta_0) * lse
            return self._like(tmp / (np.sqrt(self.beta_0 ** 2 + tmp) + self.beta_0) / (self.beta_1 - self.beta_0), lamb)
        if self.schedule == "cosine":        # uni_pc.py:145-149
            return self._like(np.arccos(np.exp(-0.5 * lse + self._cos_la0)) * 2.0 * (1.0 + self._COS_S) / np.pi - self._COS_S, lamb)
        la = -0.5 * lse
        return self._like(self._interp(la, self._yp[::-1], self._xp[::-1]), lamb)

    def _plan_schedule(self):
        """(kind, beta_0, beta_1) for the native plan (dv_sampler_plan_sched)."""
        return (self.schedule, self.beta_0, self.beta_1)


def wrap_model(model, noise_schedule, model_type="noise", model_kwargs={}, guidance_type="uncond", condition=None,
               unconditional_condition=None, guidance_scale=1.0, classifier_fn=None, classifier_kwargs={}):
    """`model_wrapper(...)` (reference sampler/dpm_solver.py:170-334): returns the
    noise-prediction function `model_fn(x, t_continuous)` the solvers take.  The closure also
    carries the raw model and its kwargs so the solver can evaluate the x0-prediction
    directly instead of converting x0 -> noise -> x0."""
    if model_type not in ("noise", "x_start", "v", "score"):
        raise AssertionError("model_type must be one of noise / x_start / v / score")
    if guidance_type not in ("uncond", "classifier", "classifier-free"):
        raise AssertionError("guidance_type must be one of uncond / classifier / classifier-free")
    ns = noise_schedule

    def expand(v, x):
        return v.reshape((-1,) + (1,) * (x.dim() - 1))

    def t_input(t_continuous):       # get_model_input_time (dpm_solver.py:271-280)
        if ns.schedule != "discrete":
            return t_continuous
        return (t_continuous - 1.0 / ns.total_N) * ns.total_N

    def noise_pred_fn(x, t_continuous, cond=None):     # dpm_solver.py:282-298
        if cond is None:
            out = model(x, t_input(t_continuous), **model_kwargs)
        else:
            out = model(x, t_input(t_continuous), cond, **model_kwargs)
        if model_type == "noise":
            return out
        alpha_t = ns.marginal_alpha(t_continuous).to(x)
        sigma_t = ns.marginal_std(t_continuous).to(x)
        if model_type == "x_start":
            return (x - expand(alpha_t, x) * out) / expand(sigma_t, x)
        if model_type == "v":
            return expand(alpha_t, x) * out + expand(sigma_t, x) * x
        return -expand(sigma_t, x) * out

    def cond_grad_fn(x, t_in):                         # grad_x log p_t(condition | x_t), dpm_solver.py:300-307
        with torch.enable_grad():
            x_in = x.detach().requires_grad_(True)
            log_prob = classifier_fn(x_in, t_in, condition, **classifier_kwargs)
            return torch.autograd.grad(log_prob.sum(), x_in)[0]

    def model_fn(x, t_continuous):                     # dpm_solver.py:309-330
        if guidance_type == "uncond":
            return noise_pred_fn(x, t_continuous)
        if guidance_type == "classifier":
            assert classifier_fn is not None
            cond_grad = cond_grad_fn(x, t_input(t_continuous))
            sigma_t = ns.marginal_std(t_continuous).to(x)
            return noise_pred_fn(x, t_continuous) - guidance_scale * expand(sigma_t, x) * cond_grad
        if guidance_scale == 1.0 or unconditional_condition is None:
            return noise_pred_fn(x, t_continuous, cond=condition)
        x_in = torch.cat([x] * 2)
        t_in = torch.cat([t_continuous] * 2)
        c_in = torch.cat([unconditional_condition, condition])
        noise_uncond, noise = noise_pred_fn(x_in, t_in, cond=c_in).chunk(2)
        return noise_uncond + guidance_scale * (noise - noise_uncond)

    # (the direct x0 route - and with it the native graph - is taken for the unguided wrapper only: a guided model_fn is an
    # arbitrary noise prediction for the solvers)
    if guidance_type == "uncond":
        model_fn._dv = dict(model=model, model_type=model_type, model_kwargs=model_kwargs, noise_schedule=ns)
    return model_fn


class Plan:
    """Compiled multistep loop: events + fp64-derived coefficient rows."""

    def __init__(self, solver, betas, steps, order, skip_type, lower_order_final, t_start=None, t_end=None,
                 denoise_to_zero=False, schedule=("discrete", 0.0, 0.0), method="multistep"):
        L = _lib()
        self.method = method
        for name, v in (("t_start", t_start), ("t_end", t_end)):
            # reference dpm_solver.py:1159 / uni_pc.py:598
            assert v is None or v > 0, ("Time range needs to be greater than 0. For discrete-time DPMs, it needs to be in "
                                        "[1 / N, 1], where N is the length of betas array")
        if skip_type not in L.SKIP:
            raise ValueError("Unsupported skip_type {}, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'"
                             .format(skip_type))
        betas = np.ascontiguousarray(betas, dtype=np.float32)
        self._args = (solver, betas, steps, order, skip_type, lower_order_final, t_start, t_end, denoise_to_zero, schedule, method)
        self._per_shape = {}          # captured graphs live in the native plan, one per plan: a copy per input shape
        self._h = C.c_void_p()
        L.check(L.lib().dv_sampler_plan_method(solver, L.SCHEDULE[schedule[0]], betas.ctypes.data_as(C.c_void_p), len(betas),
                                               float(schedule[1]), float(schedule[2]), L.METHOD[method], steps, order,
                                               L.SKIP[skip_type], int(bool(lower_order_final)),
                                               -1.0 if t_start is None else float(t_start), -1.0 if t_end is None else float(t_end),
                                               int(bool(denoise_to_zero)), C.byref(self._h)),
                "dv_sampler_plan_method")
        nfe = C.c_int32()
        L.check(L.lib().dv_plan_info(self._h, C.byref(nfe), None, None), "dv_plan_info")
        self.nfe = nfe.value
        nts = C.c_int32()
        L.check(L.lib().dv_plan_times(self._h, C.byref(nts), None), "dv_plan_times")
        self.t_input = np.zeros(self.nfe, dtype=np.float64)
        self.timesteps = np.zeros(nts.value, dtype=np.float64)     # steps + 1 points (singlestep methods: outer steps + 1)
        self.eval_times = np.zeros(self.nfe, dtype=np.float64)     # continuous time of every evaluation
        L.check(L.lib().dv_plan_info(self._h, None, self.t_input.ctypes.data_as(C.c_void_p),
                                     self.timesteps.ctypes.data_as(C.c_void_p)), "dv_plan_info")
        L.check(L.lib().dv_plan_times(self._h, None, self.eval_times.ctypes.data_as(C.c_void_p)), "dv_plan_times")
        n = C.c_int32()
        L.check(L.lib().dv_plan_coefs(self._h, C.byref(n), None), "dv_plan_coefs")
        self.coefs = np.zeros((n.value, 8), dtype=np.float32)
        L.check(L.lib().dv_plan_coefs(self._h, None, self.coefs.ctypes.data_as(C.c_void_p)), "dv_plan_coefs")
        ne, ns = C.c_int32(), C.c_int32()
        L.check(L.lib().dv_plan_events(self._h, C.byref(ne), None, C.byref(ns)), "dv_plan_events")
        self.events = np.zeros((ne.value, 9), dtype=np.int32)
        self.n_slots = ns.value
        L.check(L.lib().dv_plan_events(self._h, None, self.events.ctypes.data_as(C.c_void_p), None), "dv_plan_events")

    def for_shape(self, shape):
        """The plan whose captured hipGraph belongs to inputs of this shape (this one for the first shape seen, a copy
        for every further one: utterances of alternating lengths then replay their graphs instead of re-capturing)."""
        if not self._per_shape:
            self._per_shape[shape] = self
        if shape not in self._per_shape:
            if len(self._per_shape) >= 8:                      # bounded: drop the oldest copy
                old = next(k for k in self._per_shape if self._per_shape[k] is not self)
                del self._per_shape[old]
            self._per_shape[shape] = Plan(*self._args)
        return self._per_shape[shape]

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib().lib().dv_plan_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def run_python(self, x, data_model, intermediates=None, x0_hook=None, xt_hook=None):
        """Execute the loop with torch ops; `data_model(x, eval_idx)` returns the x0 prediction.  `intermediates`
        (a list) receives what the reference's return_intermediate collects: the start point, x after every step
        and, with denoise_to_zero, the final data prediction (dpm_solver.py:1179-1240).
        `x0_hook(x0, eval_idx)` corrects a data prediction (correcting_x0_fn / dynamic thresholding, dpm_solver.py:433-445);
        `xt_hook(x, t, step)` corrects the state after the first evaluation (step 0) and after every step
        (correcting_xt_fn, dpm_solver.py:1180-1181, 1188-1189, 1203-1204, 1237-1238; uni_pc.py:615-647, 664-665)."""
        hist = [None] * self.n_slots
        xp = None
        start_pending = False          # the start point: recorded (and corrected) once the first evaluation is complete
        step = 0
        t_of = lambda k: torch.tensor(float(self.timesteps[min(k, len(self.timesteps) - 1)]), dtype=torch.float32, device=x.device)

        # the singlestep methods neither record nor correct the start point, and number their steps from 0
        # (dpm_solver.py:1221-1232 against :1179-1183)
        single = self.method != "multistep"

        def finish_start(x):
            if single:
                return x
            if xt_hook is not None:
                x = xt_hook(x, t_of(0), 0)
            if intermediates is not None:
                intermediates.append(x)
            return x

        seen_eval = False
        for typ, src, eidx, dst, coef, s0, s1, s2, s3 in self.events.tolist():
            if start_pending and not (typ == 1 and dst >= 2):      # (an x0 -> noise conversion still reads the uncorrected x)
                x = finish_start(x)
                start_pending = False
            if typ == 0:
                m = data_model(x if src == 0 else xp, eidx)
                hist[dst] = m if x0_hook is None else x0_hook(m, eidx)
                if not seen_eval:
                    start_pending = True
                seen_eval = True
            else:
                c = self.coefs[coef]
                if float(c[7]) != 0.0:     # x0 -> noise prediction: (x - alpha m) / sigma, the reference's order of operations (sampler.hip)
                    out = ((x if src == 0 else xp) - float(c[1]) * hist[s0]) / float(c[0])
                else:
                    out = float(c[0]) * (x if src == 0 else xp)      # src 1: continuation of a sum chained through x_pred
                    for k, s in enumerate((s0, s1, s2, s3)):
                        if s >= 0:
                            out = out + float(c[1 + k]) * hist[s]
                if dst == 0:
                    x = out
                    step += 1
                    if xt_hook is not None:     # (singlestep: the steps - and the final denoise step behind them - count from 0)
                        x = xt_hook(x, t_of(step), step - 1 if single else step)
                    if intermediates is not None:
                        intermediates.append(x)
                elif dst == 1:
                    xp = out
                else:                      # a history slot in place (x0 -> noise prediction: algorithm_type='dpmsolver')
                    hist[dst - 2] = out
        if start_pending:
            x = finish_start(x)
        return x




class NativeUNetModel:
    """Opt-in marker for the fully native loop: pass an instance as `model` to `model_wrapper`
    (model_type='x_start').  It is also a plain callable `(x, t_input) -> x0`, so every other
    solver path keeps working.  Holds this package's UNet (backend='hip'), the channel-concat
    condition `cond` [B, in-out, T] and the cross-attention inputs."""

    def __init__(self, unet, cond, encoder_hidden_states, encoder_attention_mask=None):
        self.unet, self.cond, self.enc, self.mask = unet, cond, encoder_hidden_states, encoder_attention_mask

    def __call__(self, x, t_input, **kwargs):
        sample = x if self.cond is None else torch.cat([x, self.cond], dim=1)
        return self.unet(sample, t_input, self.enc, encoder_attention_mask=self.mask).sample

    def run_plan(self, plan, x):
        """hipGraph replay of the whole loop (dv_sampler_run).  Returns a new tensor.

        Does not block the host in the steady state (engine.verify_handover_default(): "lazy").  While the engine's schedule
        uses in-launch hand-overs (the default on a GPU the process has to itself) the first run of a newly planned schedule is
        verified before its result is returned - wait for the stream, read the flag, repeat on the fallback schedule if a wait
        timed out; later runs are asynchronous: a time-out is reported by the next call on the engine (RuntimeError: repeat the
        run) or by `unet.hip_engine().wait()` (False).  DVITS_HANDOVER_VERIFY=eager restores one host wait per run."""
        L = _lib()
        eng = self.unet.hip_engine()
        out = self._run_plan_once(plan, x, eng, L)
        # In-launch hand-overs need the device to themselves; on a shared GPU one may time out (bounded wait, flag in host
        # memory): the engine then drops to the fallback schedule and the run is repeated on it - degraded, not dead
        # (VERDICT r2 #6); when that check happens: UNetEngine.result_leaves
        out2 = eng.result_leaves(lambda: self._run_plan_once(plan, x, eng, L))
        return out if out2 is None else out2

    def _run_plan_once(self, plan, x, eng, L):
        B, C_, T = x.shape
        eng.before_enqueue()             # (lazy verification: a run before a long host pause is verified here - engine.py)
        eng.prepare(B, T, self.enc.shape[1])
        bias = self.unet._bias_from_mask(self.mask, torch.float32)
        eng.set_cond(self.enc, bias)
        # Persistent input / condition buffers per shape: the captured graph is keyed on their addresses (and on the
        # engine's schedule handle), so a new utterance of a shape seen before replays its graph instead of re-capturing
        # steps x ~160 launches.  (Bounded: the engine keeps DVITS_PLAN_CACHE schedules.)
        bufs = self.__dict__.setdefault("_bufs", {})
        key = (tuple(x.shape), None if self.cond is None else tuple(self.cond.shape), str(x.device))
        if key not in bufs:
            if len(bufs) >= 8:
                bufs.pop(next(iter(bufs)))
            bufs[key] = (torch.empty(x.shape, device=x.device, dtype=torch.float32),
                         None if self.cond is None else torch.empty(self.cond.shape, device=self.cond.device, dtype=torch.float32))
        xbuf, cbuf = bufs[key]
        xbuf.copy_(x)
        if cbuf is not None:
            cbuf.copy_(self.cond)
        plan = plan.for_shape((key, eng.handle.value))
        eng.check_or_recover(L.lib().dv_sampler_run(plan.handle, eng.handle, L.ptr(xbuf), L.ptr(cbuf), L.stream_ptr()),
                             "dv_sampler_run")
        return xbuf.clone()


def dynamic_thresholding(x0, ratio, max_val):
    """dynamic_thresholding_fn (dpm_solver.py:416-425, uni_pc.py:268-277): per sample, s = max(quantile(|x0|, ratio), max_val);
    x0 <- clamp(x0, -s, s) / s."""
    s = torch.quantile(torch.abs(x0).reshape((x0.shape[0], -1)), ratio, dim=1)
    s = torch.maximum(s, max_val * torch.ones_like(s)).reshape((-1,) + (1,) * (x0.dim() - 1))
    return torch.clamp(x0, -s, s) / s


def sample_with_plan(plan, model_fn, noise_schedule, x, intermediates=None, x0_hook=None, xt_hook=None):
    """Run a compiled loop for a solver-level `model_fn` (noise prediction, as returned by
    model_wrapper or supplied by the user).  With `intermediates` (a list to fill) or a correction hook the loop runs step
    by step from Python (same kernels per evaluation) instead of as one graph replay."""
    info = getattr(model_fn, "_dv", None)
    ns = noise_schedule
    B = x.shape[0]
    with torch.no_grad():
        if info is not None and info["model_type"] == "x_start":
            raw = info["model"]
            if (isinstance(raw, NativeUNetModel) and x.is_cuda and raw.unet.backend == "hip" and not info["model_kwargs"]
                    and intermediates is None and x0_hook is None and xt_hook is None):
                return raw.run_plan(plan, x)
            kwargs = info["model_kwargs"]

            def data_model(xx, eidx):
                t_in = torch.full((B,), float(plan.t_input[eidx]), device=xx.device, dtype=torch.float32)
                return raw(xx, t_in, **kwargs)
        else:
            # arbitrary noise-prediction callable: x0 = (x - sigma_t * eps) / alpha_t
            # (reference data_prediction_fn, dpm_solver.py:433-442)
            eval_t = _eval_times(plan)

            def data_model(xx, eidx):
                t = torch.full((B,), float(eval_t[eidx]), device=xx.device, dtype=torch.float32)
                noise = model_fn(xx, t)
                a = float(ns.marginal_alpha(torch.tensor([eval_t[eidx]], dtype=torch.float64))[0])
                s = float(ns.marginal_std(torch.tensor([eval_t[eidx]], dtype=torch.float64))[0])
                return (xx - s * noise) / a
        return plan.run_python(x, data_model, intermediates, x0_hook, xt_hook)


def _eval_times(plan):
    """Continuous time of each model evaluation (inverse of t_input = (t - 1/N) * N is not
    needed: evaluations happen at the grid points in order; UniPC evaluates at the step's
    end time, DPM-Solver++ at the step's start)."""
    # (multistep: EVAL k is at timesteps[k]; the singlestep methods evaluate inside their outer steps)
    return plan.eval_times
