#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (build container only).

The reference has no tests or golden vectors of its own for the sampling path
(SURVEY.md §4), so every pin comes from running the reference itself here on seeded
synthetic weights/inputs (diff_vits_amd.synth) and committing the small outputs.
Only inputs' seeds/configs and expected outputs are stored: no reference source, no
weights.  Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--ref /root/reference]

It also prints how far oracle/ and the package's torch backend are from the reference on
the same inputs (expected: float32 rounding).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import diff_vits_amd  # noqa: E402,F401
from diff_vits_amd import synth  # noqa: E402
from diff_vits_amd.unet1d.unet_1d_condition import UNet1DConditionModel as OurUNet  # noqa: E402
from oracle import sampler_ref, unet_ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

UNET_CASES = {
    # name: (ctor kwargs, B, T, L, ragged mask, timestep spec)
    "tiny": (dict(in_channels=24, out_channels=8, block_out_channels=(32, 64, 96, 128), norm_num_groups=8,
                  cross_attention_dim=32, attention_head_dim=8, addition_embed_type="text",
                  resnet_time_scale_shift="scale_shift", addition_embed_type_num_heads=8), 2, 40, 12, True, "frac"),
    "cfg1": (dict(in_channels=208, out_channels=80, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
                  cross_attention_dim=128, attention_head_dim=8, addition_embed_type="text",
                  resnet_time_scale_shift="scale_shift"), 1, 256, 128, False, "frac"),
    "oddT": (dict(in_channels=208, out_channels=80, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
                  cross_attention_dim=128, attention_head_dim=8, addition_embed_type="text",
                  resnet_time_scale_shift="scale_shift"), 2, 100, 50, True, "frac"),
    "c100": (dict(in_channels=228, out_channels=100, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
                  cross_attention_dim=128, attention_head_dim=8, addition_embed_type="text",
                  resnet_time_scale_shift="scale_shift"), 2, 64, 40, False, "frac"),
    "durpred": (dict(in_channels=256, out_channels=1, block_out_channels=(64, 64, 128, 128), norm_num_groups=8,
                     cross_attention_dim=256, attention_head_dim=8, addition_embed_type="text",
                     resnet_time_scale_shift="scale_shift"), 2, 37, 60, True, "int1"),
}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def oracle_cfg(kw):
    return unet_ref.default_config(kw["in_channels"], kw["out_channels"], kw["block_out_channels"],
                                   kw["cross_attention_dim"], kw["attention_head_dim"], kw["norm_num_groups"], 2,
                                   kw.get("addition_embed_type_num_heads", 64))


def unet_inputs(kw, B, T, L, ragged, tspec, seed=1234):
    cin = kw["in_channels"]
    sample = synth.normal(seed, "sample", (B, cin, T))
    enc = synth.normal(seed, "enc", (B, L, kw["cross_attention_dim"]))
    mask = np.ones((B, L), dtype=bool)
    if ragged:
        for b in range(B):
            mask[b, max(1, L - 5 * (b + 1)):] = False
    if tspec == "frac":
        t = np.array([949.05 - 37.5 * b for b in range(B)], dtype=np.float32)
    else:
        t = 1
    return sample, t, enc, mask


def probe_reduce(p):
    """Keep goldens small: store each probe's per-(batch,channel) mean and the first 8 frames."""
    out = {}
    for k, v in p.items():
        v = v.detach().numpy()
        if v.ndim == 3:
            out["probe_mean_" + k] = v.mean(axis=2)
            out["probe_head_" + k] = v[:, :, :8].copy()
        else:
            out["probe_" + k] = v
    return out


def make_unet(ref_unet_cls, name):
    kw, B, T, L, ragged, tspec = UNET_CASES[name]
    torch.manual_seed(0)
    ref = ref_unet_cls(**kw).eval()
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    with torch.device("meta"):
        ours_meta = OurUNet(**kw)
    our_shapes = {k: tuple(v.shape) for k, v in ours_meta.state_dict().items()}
    assert shapes == our_shapes, ("state-dict layout differs from the reference", set(shapes.items()) ^ set(our_shapes.items()))
    sd_np = synth.make_state_dict(shapes, seed=1234)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    ref.load_state_dict(sd)
    sample, t, enc, mask = unet_inputs(kw, B, T, L, ragged, tspec)
    ts = torch.from_numpy(t) if isinstance(t, np.ndarray) else t
    if name == "durpred":  # float [B,1,L] mask as reference model3.py:310,316
        m_in = torch.from_numpy(mask[:, None, :].astype(np.float32))
    else:
        m_in = torch.from_numpy(mask)
    with torch.no_grad():
        y_ref = ref(torch.from_numpy(sample), ts, torch.from_numpy(enc), encoder_attention_mask=m_in).sample
        probes = {}
        y_or = unet_ref.unet_forward(sd, oracle_cfg(kw), torch.from_numpy(sample), ts, torch.from_numpy(enc), m_in,
                                     probes=probes)
        ours = OurUNet(backend="torch", **kw).eval()
        ours.load_state_dict(sd)
        y_tb = ours(torch.from_numpy(sample), ts, torch.from_numpy(enc), encoder_attention_mask=m_in).sample
        # fp64 run of the oracle: the noise floor of the fp32 reference
        sd64 = {k: v.double() for k, v in sd.items()}
        y64 = unet_ref.unet_forward(sd64, oracle_cfg(kw), torch.from_numpy(sample).double(), ts,
                                    torch.from_numpy(enc).double(), m_in)
    print("[unet %-7s] oracle-vs-ref %.2e  torch-backend-vs-ref %.2e  ref-vs-fp64 %.2e  |y| std %.3f" % (
        name, rel(y_or, y_ref), rel(y_tb, y_ref), rel(y_ref, y64), float(y_ref.std())))
    out = dict(y=y_ref.numpy(), y64=y64.numpy().astype(np.float64) if y64.numel() <= 40000 else np.zeros(0),
               B=B, T=T, L=L, ragged=ragged, tspec=tspec, seed=1234)
    out.update(probe_reduce(probes) if name in ("tiny", "cfg1") else {})
    np.savez_compressed(os.path.join(GOLD, "unet_%s.npz" % name), **out)
    return ref, sd


def make_samplers(ref_dpm, ref_unipc, ref_unet_cls):
    betas = torch.from_numpy(synth.make_betas())
    out = {}
    # --- schedule known answers (interpolate_fn interior / edge / out of range)
    ns_d = ref_dpm.NoiseScheduleVP("discrete", betas=betas)
    ns_u = ref_unipc.NoiseScheduleVP("discrete", betas=betas)
    tq = torch.tensor([1.0, 0.9995, 0.5, 0.25005, 0.0015, 0.001, 0.0005, 1.2], dtype=torch.float32)
    out["sched_t"] = tq.numpy()
    out["sched_dpm_log_alpha"] = ns_d.marginal_log_mean_coeff(tq).numpy()
    out["sched_dpm_lambda"] = ns_d.marginal_lambda(tq).numpy()
    out["sched_dpm_std"] = ns_d.marginal_std(tq).numpy()
    out["sched_unipc_lambda"] = ns_u.marginal_lambda(tq).numpy()
    out["sched_total_N"] = np.array([ns_d.total_N, ns_u.total_N])
    lam = torch.linspace(-5.0, 4.5, 7)
    out["sched_inv_lambda_in"] = lam.numpy()
    out["sched_inv_lambda"] = ns_d.inverse_lambda(lam).numpy()

    # --- stand-in model runs
    calls = []

    def standin(x, t_input, **kw):
        calls.append(t_input.detach().clone())
        return sampler_ref.standin_model(x, t_input)

    xB = torch.from_numpy(synth.normal(7, "x_sampler", (2, 4, 16)))
    x1 = xB[:1].clone()
    dpm_cases = [(10, 2, "time_uniform"), (20, 2, "time_uniform"), (50, 2, "time_uniform"), (8, 2, "time_uniform"),
                 (20, 3, "time_uniform"), (15, 2, "logSNR"), (12, 2, "time_quadratic"), (20, 1, "time_uniform")]
    for steps, order, skip in dpm_cases:
        calls.clear()
        fn = ref_dpm.model_wrapper(standin, ns_d, model_type="x_start")
        xr, inter = ref_dpm.DPM_Solver(fn, ns_d, algorithm_type="dpmsolver++").sample(
            xB.clone(), steps=steps, order=order, skip_type=skip, method="multistep", return_intermediate=True)
        key = "dpm_s%d_o%d_%s" % (steps, order, skip)
        out[key + "_x"] = xr.numpy()
        out[key + "_x1"] = inter[1].numpy()       # after step 1 (inter[0] is the start point)
        out[key + "_tin"] = torch.stack([c[0] for c in calls]).numpy()
        xo = sampler_ref.dpm_solver_pp_sample(sampler_ref.standin_model, betas, xB.clone(), steps, order, skip)
        print("[dpm++  s=%-2d o=%d %-14s] oracle-vs-ref %.2e  NFE %d" % (steps, order, skip, rel(xo, xr), len(calls)))
    uni_cases = [(10, 2, "bh2"), (20, 2, "bh2"), (30, 2, "bh2"), (20, 2, "bh1"), (20, 3, "bh2"), (5, 2, "bh2"),
                 (20, 1, "bh2"),
                 # round 2: the general-order linear solves (uni_pc.py:545-560) and variant='vary_coeff' (:368-469)
                 (20, 4, "bh2"), (12, 5, "bh1"), (20, 6, "bh2"), (20, 1, "vary_coeff"), (20, 2, "vary_coeff"),
                 (20, 3, "vary_coeff"), (12, 4, "vary_coeff"), (6, 3, "vary_coeff")]
    for steps, order, variant in uni_cases:
        calls.clear()
        fn = ref_unipc.model_wrapper(standin, ns_u, model_type="x_start")
        xr = ref_unipc.UniPC(fn, ns_u, variant=variant).sample(
            x1.clone(), steps=steps, order=order, skip_type="time_uniform", method="multistep")
        key = "unipc_s%d_o%d_%s" % (steps, order, variant)
        out[key + "_x"] = xr.numpy()
        out[key + "_tin"] = torch.stack([c[0] for c in calls]).numpy()
        xo = sampler_ref.unipc_sample(sampler_ref.standin_model, betas, x1.clone(), steps, order, "time_uniform", variant)
        print("[unipc  s=%-2d o=%d %-4s] oracle-vs-ref %.2e  NFE %d" % (steps, order, variant, rel(xo, xr), len(calls)))
    out["x_sampler"] = xB.numpy()
    np.savez_compressed(os.path.join(GOLD, "sampler_standin.npz"), **out)

    # --- real UNet, BASELINE config 1: B=1, C=80, T=256, L=128, 20 steps
    kw = UNET_CASES["cfg1"][0]
    ref = ref_unet_cls(**kw).eval()
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(shapes, seed=1234).items()}
    ref.load_state_dict(sd)
    x, cond, enc, mask = synth.make_inputs(1, 80, 256, 128, seed=1234)
    x, cond, enc, mask = map(torch.from_numpy, (x, cond, enc, mask))

    def ref_model(xx, t_input, **kwargs):
        return ref(torch.cat([xx, cond], dim=1), t_input, enc, encoder_attention_mask=mask).sample

    o_model = unet_ref.diffusion_model_fn(sd, oracle_cfg(kw), cond, enc, mask)
    res = {}
    with torch.no_grad():
        fn = ref_dpm.model_wrapper(ref_model, ns_d, model_type="x_start")
        xr = ref_dpm.DPM_Solver(fn, ns_d, algorithm_type="dpmsolver++").sample(
            x.clone(), steps=20, order=2, skip_type="time_uniform", method="multistep")
        xo = sampler_ref.dpm_solver_pp_sample(o_model, betas, x.clone(), 20, 2)
        print("[cfg1 dpm++ 20 steps real UNet] oracle-vs-ref %.2e" % rel(xo, xr))
        res["dpm_x"] = xr.numpy()
        fn = ref_unipc.model_wrapper(ref_model, ns_u, model_type="x_start")
        xr = ref_unipc.UniPC(fn, ns_u, variant="bh2").sample(
            x.clone(), steps=20, order=2, skip_type="time_uniform", method="multistep")
        xo = sampler_ref.unipc_sample(o_model, betas, x.clone(), 20, 2)
        print("[cfg1 unipc 20 steps real UNet] oracle-vs-ref %.2e" % rel(xo, xr))
        res["unipc_x"] = xr.numpy()
    np.savez_compressed(os.path.join(GOLD, "sampler_cfg1.npz"), **res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    sys.path.insert(1, args.ref)
    sys.dont_write_bytecode = True
    # the reference's `unet1d` / `sampler` top-level packages
    import importlib
    ref_unet = importlib.import_module("unet1d.unet_1d_condition")
    assert ref_unet.__file__.startswith(args.ref), ref_unet.__file__
    ref_dpm = importlib.import_module("sampler.dpm_solver")
    ref_unipc = importlib.import_module("sampler.uni_pc")
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    for name in UNET_CASES:
        if args.only and name not in args.only:
            continue
        make_unet(ref_unet.UNet1DConditionModel, name)
    if not args.only or "sampler" in args.only:
        make_samplers(ref_dpm, ref_unipc, ref_unet.UNet1DConditionModel)


if __name__ == "__main__":
    main()
