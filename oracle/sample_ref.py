"""CPU restatement of the reference's sampling orchestration, NaturalSpeech2.sample (model3.py:1118-1203) with the
schedule buffers of NaturalSpeech2.__init__ (model3.py:976-1006).

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline).  Pinned by tools/make_golden_prompt.py against
the stub-imported reference with the prior (`vits.infer`), `torch.randn` and the vocoder replaced by fixed stand-ins:
tests/golden/sample_unipc.npz.  The reference's 'dpmsolver' branch cannot run as written (`vits.infer`
is called with a tuple, model3.py:1138-1140); `sample_method="dpmsolver"` here uses the 'unipc' branch's plumbing
with the solver call of the 'dpmsolver' branch (40 steps, order 2, time_uniform, multistep, model3.py:1148-1158).
"""
import torch
import torch.nn.functional as F

from . import prompt_ref, sampler_ref, unet_ref


def schedule_buffers(timesteps=1000):
    """model3.py:935-942, 976-1006: float64 linear betas and derived tables, stored as float32 buffers."""
    scale = 1000 / timesteps
    betas = torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)
    alphas = 1. - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.)
    pv = betas * (1. - ac_prev) / (1. - ac)
    out = {
        "betas": betas, "alphas_cumprod": ac, "alphas_cumprod_prev": ac_prev, "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1. - ac), "log_one_minus_alphas_cumprod": torch.log(1. - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1. / ac), "sqrt_recipm1_alphas_cumprod": torch.sqrt(1. / ac - 1),
        "posterior_variance": pv, "posterior_log_variance_clipped": torch.log(pv.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1. - ac),
        "posterior_mean_coef2": (1. - ac_prev) * torch.sqrt(alphas) / (1. - ac),
    }
    return {k: v.to(torch.float32) for k, v in out.items()}


def sample_mel(diff_sd, dcfg, content, refer, text_lengths, spec_lengths, noise, sample_method="unipc", timesteps=1000):
    """(content, refer) from the prior -> mel.  diff_sd: Diffusion_Encoder state dict ('unet.*', 'prompt_encoder.*')."""
    H = dcfg["hidden_channels"]
    ucfg = unet_ref.default_config(dcfg["in_channels"] + H, dcfg["out_channels"], (128, 256, 384, 512), H, dcfg["n_heads"], 8, 2, 64)
    betas = schedule_buffers(timesteps)["betas"]

    def model(x, t_input):          # sample_fun (model3.py:1113-1118): x_start = diff_model(x, data, t)
        return prompt_ref.diffusion_encoder_forward(diff_sd, ucfg, x, content, refer, spec_lengths, t_input)

    with torch.no_grad():
        if sample_method == "unipc":
            return sampler_ref.unipc_sample(model, betas, noise, steps=30, order=2, skip_type="time_uniform", variant="bh2")
        if sample_method == "dpmsolver":
            return sampler_ref.dpm_solver_pp_sample(model, betas, noise, steps=40, order=2, skip_type="time_uniform")
    raise ValueError("sample_method %r is not on the supported path" % (sample_method,))
