"""DPM-Solver++ mirror of the reference `sampler/dpm_solver.py` (public symbols
`NoiseScheduleVP`, `model_wrapper`, `DPM_Solver`; reference model3.py:1128 imports exactly these).

`DPM_Solver.sample(..., method='multistep')` compiles the loop once on the host (fp64 tables,
libdvits_hip.so `dv_sampler_plan`) and then either replays it natively (hipGraph: UNet schedule
+ fused update kernels) when the wrapped model is a `NativeUNetModel`, or runs it with torch ops
around an arbitrary Python callable.  Scope of this build: algorithm_type='dpmsolver++' and 'dpmsolver',
methods 'multistep' (reference :1171-1213, :547-592, :796-904), 'singlestep' and 'singlestep_fixed' (:482-539, :594-794,
:1214-1232) - compiled - and 'adaptive' (:906-1010, stepped from Python: sampler/_adaptive.py), orders 1-3, solver types 'dpmsolver' and 'taylor', schedules 'discrete' and 'linear'.
"""
import torch

from ._adaptive import adaptive_sample
from ._plan import NativeUNetModel, NoiseScheduleBase, Plan, _eval_times, dynamic_thresholding, sample_with_plan, wrap_model

__all__ = ["NoiseScheduleVP", "model_wrapper", "DPM_Solver", "NativeUNetModel"]

_SOLVER_DPMPP = 0
_SOLVER_DPM = 4          # algorithm_type='dpmsolver': multistep updates on the noise prediction (include/dvits_hip.h)
_TAYLOR = {_SOLVER_DPMPP: 5, _SOLVER_DPM: 6}     # solver_type='taylor': the second-order update's Taylor form


class NoiseScheduleVP(NoiseScheduleBase):
    """VP schedule: 'discrete' with the log-SNR clip at -5.1, or the continuous-time 'linear' (reference dpm_solver.py:6-167)."""
    clip_lambda = -5.1
    schedules = ("discrete", "linear")


model_wrapper = wrap_model


class DPM_Solver:
    def __init__(self, model_fn, noise_schedule, algorithm_type="dpmsolver++", correcting_x0_fn=None,
                 correcting_xt_fn=None, thresholding_max_val=1.0, dynamic_thresholding_ratio=0.995):
        assert algorithm_type in ["dpmsolver", "dpmsolver++"]
        # correcting_x0_fn ("dynamic_thresholding" or fn(x0, t)) / correcting_xt_fn (fn(x, t, step)): reference :409-415.  With
        # either the loop is stepped from Python (the hooks are arbitrary callables), never replayed as one graph.
        if correcting_x0_fn == "dynamic_thresholding":
            correcting_x0_fn = lambda x0, t: dynamic_thresholding(x0, dynamic_thresholding_ratio, thresholding_max_val)
        self.correcting_x0_fn, self.correcting_xt_fn = correcting_x0_fn, correcting_xt_fn
        self.model_fn = model_fn
        self.noise_schedule = noise_schedule
        self.algorithm_type = algorithm_type
        self._plans = {}

    def get_time_steps(self, skip_type, t_T, t_0, N, device):
        """Time grid of the loop from t_T down to t_0 (reference dpm_solver.py:453-480)."""
        plan = self._plan(N, min(2, N), skip_type, True, float(t_T), float(t_0))
        return torch.as_tensor(plan.timesteps, dtype=torch.float32, device=device)

    def _plan(self, steps, order, skip_type, lower_order_final, t_start=None, t_end=None, denoise_to_zero=False,
              solver_type="dpmsolver", method="multistep"):
        key = (steps, order, skip_type, bool(lower_order_final), t_start, t_end, bool(denoise_to_zero), solver_type, method)
        if key not in self._plans:
            solver = _SOLVER_DPMPP if self.algorithm_type == "dpmsolver++" else _SOLVER_DPM
            if solver_type == "taylor":
                solver = _TAYLOR[solver]
            self._plans[key] = Plan(solver, self.noise_schedule._betas, steps, order, skip_type,
                                    lower_order_final, t_start, t_end, denoise_to_zero,
                                    schedule=self.noise_schedule._plan_schedule(), method=method)
        return self._plans[key]

    def sample(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type="time_uniform", method="multistep",
               lower_order_final=True, denoise_to_zero=False, solver_type="dpmsolver", atol=0.0078, rtol=0.05,
               return_intermediate=False):
        """x at t_start (default T) -> x at t_end (default 1/N), reference dpm_solver.py:1047-1245.  NFE == steps
        (+1 with denoise_to_zero).  return_intermediate=True returns (x, [start point, x after every step, ...])."""
        if method not in ("multistep", "singlestep", "singlestep_fixed", "adaptive"):
            raise ValueError("Got wrong method {}".format(method))
        if method == "adaptive":
            # data-dependent step sizes: stepped from Python (sampler/_adaptive.py; reference :906-1010, :1164-1170, :1234-1240)
            assert self.correcting_xt_fn is None, "Cannot use adaptive solver when correcting_xt_fn is not None"
            assert not return_intermediate, "Cannot use adaptive solver when saving intermediate values"      # (reference :1162-1163)
            ns = self.noise_schedule
            t_0 = 1.0 / ns.total_N if t_end is None else t_end
            t_T = ns.T if t_start is None else t_start
            assert t_0 > 0 and t_T > 0, "Time range needs to be greater than 0. For discrete-time DPMs, it needs to be in [1 / N, 1], where N is the length of betas array"
            with torch.no_grad():
                x, self.last_adaptive_nfe = adaptive_sample(self, x, order, t_T, t_0, atol=atol, rtol=rtol, solver_type=solver_type)
                if denoise_to_zero:
                    saved, self.algorithm_type = self.algorithm_type, "dpmsolver++"      # (denoise_to_zero_fn = data_prediction_fn, :541-545)
                    try:
                        from ._adaptive import _prediction_fn
                        x = _prediction_fn(self)[0](x, torch.ones((1,), dtype=torch.float32, device=x.device) * t_0)
                    finally:
                        self.algorithm_type = saved
            return x
        if order not in (1, 2, 3):
            raise ValueError("Solver order must be 1 or 2 or 3, got {}".format(order))
        if solver_type not in ("dpmsolver", "taylor"):
            raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
        assert steps >= order
        plan = self._plan(steps, order, skip_type, lower_order_final, t_start, t_end, denoise_to_zero, solver_type, method)
        x0_hook = None
        if self.correcting_x0_fn is not None:
            # data_prediction_fn applies it (:433-445): every evaluation of 'dpmsolver++'; with 'dpmsolver' only the final
            # denoise_to_zero evaluation goes through data_prediction_fn (:541-545)
            times, fn0, last = _eval_times(plan), self.correcting_x0_fn, plan.nfe - 1
            only_last = self.algorithm_type != "dpmsolver++"

            def x0_hook(x0, eidx):
                if only_last and not (denoise_to_zero and eidx == last):
                    return x0
                t = torch.tensor(float(times[eidx]), dtype=torch.float32, device=x0.device)
                return fn0(x0, t)
        hooks = dict(x0_hook=x0_hook, xt_hook=self.correcting_xt_fn)
        if not return_intermediate:
            return sample_with_plan(plan, self.model_fn, self.noise_schedule, x, **hooks)
        inter = []
        out = sample_with_plan(plan, self.model_fn, self.noise_schedule, x, inter, **hooks)
        return out, inter
