"""UniPC mirror of the reference `sampler/uni_pc.py` (public symbols `NoiseScheduleVP`,
`model_wrapper`, `UniPC`; reference model3.py:1162 imports exactly these).

Same execution model as `dpm_solver.py` in this package: the multistep predictor-corrector loop - variants
'bh1' / 'bh2' (reference uni_pc.py:471-588) and 'vary_coeff' (:368-469), orders 1..8, loop :590-672 - is compiled on the host into linear-combination
coefficients and replayed natively or around a Python callable.  Unlike the reference's
'x_start' wrapper (uni_pc.py:189-191, which only broadcasts correctly at B == 1) the batch
broadcast here is the intended per-sample one; at B == 1 both coincide.
"""
from ._plan import NativeUNetModel, NoiseScheduleBase, Plan, dynamic_thresholding, sample_with_plan, wrap_model

__all__ = ["NoiseScheduleVP", "model_wrapper", "UniPC", "NativeUNetModel"]

_SOLVERS = {"bh1": 1, "bh2": 2, "vary_coeff": 3}
MAX_ORDER = 8


class NoiseScheduleVP(NoiseScheduleBase):
    """VP schedule without log-SNR clipping: 'discrete', 'linear' or 'cosine' (reference uni_pc.py:6-152)."""
    clip_lambda = None
    schedules = ("discrete", "linear", "cosine")


model_wrapper = wrap_model


class UniPC:
    def __init__(self, model_fn, noise_schedule, algorithm_type="data_prediction", correcting_x0_fn=None,
                 correcting_xt_fn=None, thresholding_max_val=1.0, dynamic_thresholding_ratio=0.995, variant="bh1"):
        assert algorithm_type in ["data_prediction", "noise_prediction"]
        # correcting_x0_fn ("dynamic_thresholding" or fn(x0)) / correcting_xt_fn (fn(x, t, step)): reference :256-261, 292-293.
        # With either the loop is stepped from Python, never replayed as one graph.
        if correcting_x0_fn == "dynamic_thresholding":
            correcting_x0_fn = lambda x0: dynamic_thresholding(x0, dynamic_thresholding_ratio, thresholding_max_val)
        self.correcting_x0_fn, self.correcting_xt_fn = correcting_x0_fn, correcting_xt_fn
        if variant not in _SOLVERS:
            raise NotImplementedError("variant %r (supported: 'bh1', 'bh2', 'vary_coeff')" % (variant,))
        self.model_fn = model_fn
        self.noise_schedule = noise_schedule
        self.variant = variant
        self.predict_x0 = algorithm_type == "data_prediction"    # (False: the same updates on the noise prediction, uni_pc.py:266)
        self._plans = {}

    def _plan(self, steps, order, skip_type, lower_order_final, t_start=None, t_end=None, denoise_to_zero=False):
        key = (steps, order, skip_type, bool(lower_order_final), t_start, t_end, bool(denoise_to_zero))
        if key not in self._plans:
            self._plans[key] = Plan(_SOLVERS[self.variant] + (0 if self.predict_x0 else 6), self.noise_schedule._betas, steps, order, skip_type,
                                    lower_order_final, t_start, t_end, denoise_to_zero,
                                    schedule=self.noise_schedule._plan_schedule())
        return self._plans[key]

    def sample(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type="time_uniform", method="multistep",
               lower_order_final=True, denoise_to_zero=False, atol=0.0078, rtol=0.05, return_intermediate=False):
        """x at t_start (default T) -> x at t_end (default 1/N), reference uni_pc.py:590-672.  NFE == steps (+1 with
        denoise_to_zero).  return_intermediate=True returns (x, [start point, x after every step, ...])."""
        if method != "multistep":
            raise ValueError("Got wrong method {}".format(method))
        if not (isinstance(order, int) and 1 <= order <= MAX_ORDER):
            raise ValueError("UniPC order must be an integer in 1..{} in this build, got {}".format(MAX_ORDER, order))
        assert steps >= order
        plan = self._plan(steps, order, skip_type, lower_order_final, t_start, t_end, denoise_to_zero)
        fn0 = self.correcting_x0_fn
        # (data_prediction_fn applies correcting_x0_fn, uni_pc.py:292-293: every evaluation of 'data_prediction'; with
        # 'noise_prediction' only the final denoise_to_zero evaluation goes through it, :279-281)
        last = plan.nfe - 1
        keep = (lambda eidx: True) if self.predict_x0 else (lambda eidx: denoise_to_zero and eidx == last)
        hooks = dict(x0_hook=None if fn0 is None else (lambda x0, eidx: fn0(x0) if keep(eidx) else x0), xt_hook=self.correcting_xt_fn)
        if not return_intermediate:
            return sample_with_plan(plan, self.model_fn, self.noise_schedule, x, **hooks)
        inter = []
        out = sample_with_plan(plan, self.model_fn, self.noise_schedule, x, inter, **hooks)
        return out, inter
