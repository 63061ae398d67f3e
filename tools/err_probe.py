#!/usr/bin/env python3
"""Relative L2 error of the HIP denoiser against the committed reference goldens (five UNet cases, the two 20-step sampler
runs of BASELINE config 1) - one line per case.  Used to weigh precision experiments (DVITS_LIB_FILE=<another build>)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from conftest import GOLD, UNET_CASES, rel_l2, unet_case  # noqa: E402
from diff_vits_amd import synth  # noqa: E402
from diff_vits_amd.sampler import dpm_solver, uni_pc  # noqa: E402
from diff_vits_amd.unet1d.unet_1d_condition import UNet1DConditionModel  # noqa: E402


def build(name):
    kw, sd, sample, t, enc, mask = unet_case(name)
    m = UNet1DConditionModel(backend="hip", **kw).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.cuda(), sample, t, enc, mask


with torch.no_grad():
    for name in UNET_CASES:
        m, sample, t, enc, mask = build(name)
        mk = torch.from_numpy(mask[:, None, :].astype(np.float32)) if name == "durpred" else torch.from_numpy(mask)
        tt = torch.from_numpy(t).cuda() if isinstance(t, np.ndarray) else t
        y = m(torch.from_numpy(sample).cuda(), tt, torch.from_numpy(enc).cuda(), encoder_attention_mask=mk.cuda()).sample
        g = np.load(os.path.join(GOLD, "unet_%s.npz" % name))["y"]
        print("unet %-8s rel_l2 %.3e   max_abs_rel %.3e" % (name, rel_l2(y.cpu().numpy(), g), float(np.abs(y.cpu().numpy() - g).max() / np.abs(g).max())))
    m, *_ = build("cfg1")
    x, cond, enc, mask = (torch.from_numpy(a).cuda() for a in synth.make_inputs(1, 80, 256, 128, seed=1234))
    betas = torch.from_numpy(synth.make_betas())
    gs = np.load(os.path.join(GOLD, "sampler_cfg1.npz"))
    for mod, key in ((dpm_solver, "dpm_x"), (uni_pc, "unipc_x")):
        ns = mod.NoiseScheduleVP("discrete", betas=betas)
        fn = mod.model_wrapper(mod.NativeUNetModel(m, cond, enc, mask), ns, model_type="x_start")
        s = mod.DPM_Solver(fn, ns, algorithm_type="dpmsolver++") if mod is dpm_solver else mod.UniPC(fn, ns, variant="bh2")
        out = s.sample(x.clone(), steps=20, order=2, skip_type="time_uniform", method="multistep")
        print("sampler %-8s rel_l2 %.3e" % (key, rel_l2(out.cpu().numpy(), gs[key])))
