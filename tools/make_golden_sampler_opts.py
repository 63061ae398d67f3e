#!/usr/bin/env python3
"""tests/golden/sampler_options.npz: the reference's DPM_Solver.sample / UniPC.sample (imported, build container only) with
the less common multistep options - custom t_start / t_end, denoise_to_zero, return_intermediate, all three skip types -
on the analytic stand-in network (x0 = tanh(x/2)(1 + 1e-6 tau)).
Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_sampler_opts.py [--ref /root/reference]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diff_vits_amd  # noqa: E402,F401
from diff_vits_amd import synth  # noqa: E402
from oracle import sampler_ref  # noqa: E402

# key: (solver, kwargs)
CASES = {
    "dpm_window": ("dpm", dict(steps=12, order=2, skip_type="time_uniform", t_start=0.8, t_end=0.05)),
    "dpm_dtz_logsnr": ("dpm", dict(steps=8, order=3, skip_type="logSNR", denoise_to_zero=True)),
    "dpm_inter_quad": ("dpm", dict(steps=10, order=2, skip_type="time_quadratic", return_intermediate=True)),
    "dpm_all": ("dpm", dict(steps=9, order=2, skip_type="time_uniform", t_start=0.95, t_end=0.01, denoise_to_zero=True,
                            return_intermediate=True)),
    "unipc_window_dtz": ("unipc", dict(steps=10, order=2, skip_type="time_uniform", t_start=0.9, t_end=0.02, denoise_to_zero=True,
                                        return_intermediate=True)),
    "unipc_o3_logsnr": ("unipc", dict(steps=7, order=3, skip_type="logSNR", t_end=0.004)),
    # continuous-time schedules (NoiseScheduleVP('linear' | 'cosine'); dpm_solver.py knows 'linear' only)
    "dpm_linear": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", schedule=("linear", 0.1, 20.0))),
    "dpm_linear_logsnr_dtz": ("dpm", dict(steps=8, order=3, skip_type="logSNR", denoise_to_zero=True, return_intermediate=True,
                                          schedule=("linear", 0.1, 20.0))),
    "unipc_linear_quad": ("unipc", dict(steps=9, order=2, skip_type="time_quadratic", t_end=0.01, schedule=("linear", 0.2, 15.0))),
    "unipc_cosine": ("unipc", dict(steps=10, order=2, skip_type="time_uniform", return_intermediate=True, schedule=("cosine", 0.1, 20.0))),
    "unipc_cosine_logsnr": ("unipc", dict(steps=8, order=3, skip_type="logSNR", schedule=("cosine", 0.1, 20.0))),
    # algorithm_type='dpmsolver': the multistep updates on the noise prediction (dpm_solver.py:581-592, 841-847, 895-904)
    "dpmn_o1": ("dpm", dict(steps=10, order=1, skip_type="time_uniform", algorithm_type="dpmsolver")),
    "dpmn_o2_dtz": ("dpm", dict(steps=12, order=2, skip_type="time_quadratic", denoise_to_zero=True, return_intermediate=True,
                                algorithm_type="dpmsolver")),
    "dpmn_o3_logsnr": ("dpm", dict(steps=8, order=3, skip_type="logSNR", algorithm_type="dpmsolver")),
    "dpmn_o3_window": ("dpm", dict(steps=20, order=3, skip_type="time_uniform", t_start=0.9, t_end=0.02, algorithm_type="dpmsolver")),
    "dpmn_linear": ("dpm", dict(steps=9, order=2, skip_type="time_uniform", schedule=("linear", 0.1, 20.0), algorithm_type="dpmsolver")),
    # correcting_x0_fn ("thr": dynamic thresholding with ratio 0.9 / max 0.6; "fn": sampler_ref.standin_x0_fix) and
    # correcting_xt_fn (sampler_ref.standin_xt_fix): dpm_solver.py:409-425, 443-444, 1180-1238; uni_pc.py:256-277, 292-293, 615-665
    "dpm_thr_xt": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", denoise_to_zero=True, return_intermediate=True,
                               hooks=("thr", True))),
    "dpm_x0fn": ("dpm", dict(steps=8, order=3, skip_type="logSNR", hooks=("fn", False))),
    "dpmn_thr_xt_dtz": ("dpm", dict(steps=9, order=2, skip_type="time_uniform", denoise_to_zero=True, algorithm_type="dpmsolver",
                                    hooks=("thr", True))),
    "unipc_thr_xt": ("unipc", dict(steps=10, order=2, skip_type="time_uniform", denoise_to_zero=True, return_intermediate=True,
                                   hooks=("thr", True))),
    "unipc_x0fn_o3": ("unipc", dict(steps=8, order=3, skip_type="time_quadratic", hooks=("fn", True))),
    # model_wrapper(guidance_type='classifier-free' | 'classifier', ...) (dpm_solver.py:282-330) on the conditional stand-in
    "dpm_cfg": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", guidance="cfg")),
    "dpm_cfg_scale1": ("dpm", dict(steps=8, order=3, skip_type="logSNR", guidance="cfg1")),
    "dpm_classifier": ("dpm", dict(steps=10, order=2, skip_type="time_quadratic", denoise_to_zero=True, guidance="clf")),
    # solver_type='taylor' (the second-order update's Taylor form, dpm_solver.py:825-829, 848-851)
    "dpm_taylor": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", solver_type="taylor")),
    "dpmn_taylor": ("dpm", dict(steps=12, order=2, skip_type="logSNR", solver_type="taylor", algorithm_type="dpmsolver")),
    # method='singlestep' ("DPM-Solver-fast": the evaluations shared out over outer steps of order <= order) and
    # 'singlestep_fixed' (dpm_solver.py:482-539, 594-794, 1214-1232)
    "dpm_ss_o3": ("dpm", dict(steps=12, order=3, skip_type="time_uniform", method="singlestep")),
    "dpm_ss_o2_logsnr_dtz": ("dpm", dict(steps=9, order=2, skip_type="logSNR", denoise_to_zero=True, return_intermediate=True,
                                         method="singlestep", hooks=(None, True))),
    "dpmn_ss_o3_quad": ("dpm", dict(steps=11, order=3, skip_type="time_quadratic", method="singlestep", algorithm_type="dpmsolver")),
    "dpm_ssfixed_taylor": ("dpm", dict(steps=12, order=3, skip_type="time_uniform", method="singlestep_fixed", solver_type="taylor")),
    "dpmn_ss_taylor_o2": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", method="singlestep", solver_type="taylor",
                                      algorithm_type="dpmsolver")),
    # UniPC(algorithm_type='noise_prediction') (uni_pc.py:266, 448-468, 569-587)
    "unipcn_bh2_o2": ("unipc", dict(steps=10, order=2, skip_type="time_uniform", unipc_algo="noise_prediction")),
    "unipcn_bh1_o3_dtz": ("unipc", dict(steps=9, order=3, skip_type="time_quadratic", denoise_to_zero=True, return_intermediate=True,
                                        unipc_algo="noise_prediction", variant="bh1", hooks=("fn", True))),
    "unipcn_vary_o4": ("unipc", dict(steps=9, order=4, skip_type="time_uniform", unipc_algo="noise_prediction", variant="vary_coeff")),
    # method='adaptive' (dpm_solver.py:906-1010; `steps` is ignored).  Step sizes follow an error estimate: where that estimate
    # is at rounding level (e.g. a first step from t_start < T on this smooth stand-in: E ~ 4e-7) the next step size amplifies the
    # rounding and two float32 / fp64 evaluations of the schedule part ways - the float32 oracle still reproduces the reference
    # bit for bit there; the cases below keep every estimate well above rounding
    "dpm_adaptive_o2": ("dpm", dict(steps=20, order=2, skip_type="time_uniform", method="adaptive")),
    "dpm_adaptive_o3_tight": ("dpm", dict(steps=20, order=3, skip_type="time_uniform", method="adaptive", atol=0.002, rtol=0.02,
                                          denoise_to_zero=True)),
    # (x_start network, order 3, loose tolerance: the accept / reject decisions sit close to E = 1 - the case that exposed the
    # skipped x0 -> noise -> x0 round trip in round 4's adaptive path)
    "dpm_adaptive_o3_loose": ("dpm", dict(steps=20, order=3, skip_type="time_uniform", method="adaptive", atol=0.01)),
    "dpmn_adaptive_o2": ("dpm", dict(steps=20, order=2, skip_type="time_uniform", method="adaptive", algorithm_type="dpmsolver")),
    "dpmn_adaptive_o3_taylor": ("dpm", dict(steps=20, order=3, skip_type="time_uniform", method="adaptive", solver_type="taylor",
                                            algorithm_type="dpmsolver", t_end=0.01)),
    # the other model types of model_wrapper (dpm_solver.py:288-298): the stand-in's output read as noise / v / score
    "dpm_type_noise": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", guidance="type:noise")),
    "dpm_type_v": ("dpm", dict(steps=10, order=3, skip_type="logSNR", guidance="type:v")),
    "dpmn_type_score": ("dpm", dict(steps=10, order=2, skip_type="time_quadratic", guidance="type:score", algorithm_type="dpmsolver")),
}


def guidance_kwargs(name, key, B):
    """model_wrapper keywords (reference and mirror) / guided_noise_fn keywords (oracle) of a case's `guidance` entry."""
    if name is None:
        return None
    if name.startswith("type:"):
        return dict(model_type=name[5:])
    cond = torch.from_numpy(synth.normal(4321, "cond." + key, (B, 5, 1)))
    if name == "clf":
        return dict(guidance_type="classifier", condition=cond, guidance_scale=1.5, classifier_fn=sampler_ref.standin_classifier)
    return dict(guidance_type="classifier-free", condition=cond, unconditional_condition=torch.zeros_like(cond),
                guidance_scale=2.5 if name == "cfg" else 1.0)


def hook_kwargs(hooks, unipc):
    """Constructor keywords of DPM_Solver / UniPC (reference or mirror) for a case's `hooks` entry."""
    if hooks is None:
        return {}
    x0, xt = hooks
    kw = {}
    if x0 == "thr":
        kw.update(correcting_x0_fn="dynamic_thresholding", dynamic_thresholding_ratio=0.9, thresholding_max_val=0.6)
    elif x0 == "fn":
        kw.update(correcting_x0_fn=(lambda v: sampler_ref.standin_x0_fix(v)) if unipc else sampler_ref.standin_x0_fix)
    if xt:
        kw.update(correcting_xt_fn=sampler_ref.standin_xt_fix)
    return kw


def oracle_hooks(hooks):
    if hooks is None:
        return {}
    x0, xt = hooks
    x0_fn = None
    if x0 == "thr":
        x0_fn = lambda v, t=None: sampler_ref.dynamic_thresholding(v, 0.9, 0.6)
    elif x0 == "fn":
        x0_fn = sampler_ref.standin_x0_fix
    return dict(x0_fn=x0_fn, xt_fn=sampler_ref.standin_xt_fix if xt else None)


def make_ns(mod, schedule, betas):
    """NoiseScheduleVP of module `mod` (reference or mirror) for a case's `schedule` entry (None: discrete)."""
    if schedule is None:
        return mod.NoiseScheduleVP("discrete", betas=betas)
    return mod.NoiseScheduleVP(schedule[0], continuous_beta_0=schedule[1], continuous_beta_1=schedule[2])


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    from sampler import dpm_solver as ref_dpm, uni_pc as ref_unipc
    torch.set_grad_enabled(False)
    betas = torch.from_numpy(synth.make_betas())
    out = {}
    for key, (solver, kw) in CASES.items():
        B = 2 if solver == "dpm" else 1                        # the reference's UniPC wrapper only broadcasts at B = 1
        x = torch.from_numpy(synth.normal(1234, "opts." + key, (B, 5, 24)))
        kw = dict(kw)
        sched = kw.pop("schedule", None)
        algo = kw.pop("algorithm_type", "dpmsolver++")
        hooks = kw.pop("hooks", None)
        method = kw.pop("method", "multistep")
        ualgo, variant = kw.pop("unipc_algo", "data_prediction"), kw.pop("variant", "bh2")
        guid = guidance_kwargs(kw.pop("guidance", None), key, B)
        net = sampler_ref.standin_cond_model if (guid and "model_type" not in guid) else sampler_ref.standin_model
        mtype = (guid or {}).get("model_type", "x_start")
        wkw = {k: v for k, v in (guid or {}).items() if k != "model_type"}
        if solver == "dpm":
            ns = make_ns(ref_dpm, sched, betas)
            fn = ref_dpm.model_wrapper(lambda xx, t, *c, **k: net(xx, t, *c), ns, model_type=mtype, **wkw)
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):      # (the adaptive solver prints its NFE)
                r = ref_dpm.DPM_Solver(fn, ns, algorithm_type=algo, **hook_kwargs(hooks, False)).sample(x.clone(), method=method, **kw)
            okw = {k: v for k, v in kw.items()}
            o = sampler_ref.dpm_solver_pp_sample(net, betas, x.clone(), okw.pop("steps"), okw.pop("order"),
                                                 okw.pop("skip_type"), schedule=sched, algorithm_type=algo, guidance=guid, method=method, **oracle_hooks(hooks), **okw)
        else:
            ns = make_ns(ref_unipc, sched, betas)
            fn = ref_unipc.model_wrapper(lambda xx, t, **k: sampler_ref.standin_model(xx, t), ns, model_type="x_start")
            r = ref_unipc.UniPC(fn, ns, variant=variant, algorithm_type=ualgo, **hook_kwargs(hooks, True)).sample(x.clone(), method="multistep", **kw)
            okw = {k: v for k, v in kw.items()}
            o = sampler_ref.unipc_sample(sampler_ref.standin_model, betas, x.clone(), okw.pop("steps"), okw.pop("order"),
                                         okw.pop("skip_type"), variant, schedule=sched, algorithm_type=ualgo, **oracle_hooks(hooks), **okw)
        if kw.get("return_intermediate"):
            xr, inter = r
            xo, ointer = o
            inter = torch.stack([t for t in inter]).numpy()     # reference: start point, every step, (denoised)
            out[key + "_inter"] = inter
            ref_steps = inter if method != "multistep" else inter[1:]   # (multistep: the oracle list omits the start point)
            assert len(ointer) == len(ref_steps)
            worst = max(rel(a.numpy(), b) for a, b in zip(ointer, ref_steps))
        else:
            xr, xo, worst = r, o, 0.0
        out[key + "_x"] = xr.numpy()
        print("%-18s oracle vs reference: final %.2e  intermediates %.2e" % (key, rel(xo.numpy(), xr.numpy()), worst))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sampler_options.npz"), **out)


if __name__ == "__main__":
    main()
