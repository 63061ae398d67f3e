#!/usr/bin/env python3
"""tests/golden/config5_b16.npz: BASELINE.json configuration 5 at reduced length, generated from the stub-IMPORTED reference
(build container only): B = 16 utterances, C = 100 mel channels (`in_channels` 228 of the shipped config.json), `cond` = the
content `VITS.infer` returns for synthetic phoneme ids (model3.py:817-860), `enc` = the prompt encoder's output for a synthetic
mel prompt (model3.py:902-914, inside `Diffusion_Encoder.forward`), 20-step DPM-Solver++ 2M through the 'unipc' branch's plumbing
(model3.py:1173-1182): `model_wrapper(self.sample_fun, ns, 'x_start', model_kwargs={'data': ...})`.

The sixteen utterances have the same text and prompt lengths; the duration predictor gives each its own frame count, so the batch
is padded to the longest by `VITS.infer` itself - every side (reference, oracle, HIP) then denoises the SAME padded batch (SURVEY
quirk 3: no length mask in the UNet's self-attention / GroupNorm).  Stored: ids, the seeds' names, frame counts and the mel.
Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_config5.py [--ref /root/reference]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import diff_vits_amd  # noqa: E402,F401
from diff_vits_amd import synth  # noqa: E402
from make_golden_prompt import import_reference, rel  # noqa: E402

B, TX, L, STEPS = 16, 48, 36, 20


def inputs(n_sym, n_tone, n_lang):
    text = torch.from_numpy((synth.uniform(1234, "cfg5.text", (B, TX)) * 0.5 + 0.5) * (n_sym - 1)).long()
    tone = torch.from_numpy((synth.uniform(1234, "cfg5.tone", (B, TX)) * 0.5 + 0.5) * (n_tone - 1)).long()
    lang = torch.from_numpy((synth.uniform(1234, "cfg5.lang", (B, TX)) * 0.5 + 0.5) * (n_lang - 1)).long()
    y = torch.from_numpy(synth.normal(1234, "cfg5.refer", (B, 100, L)))
    return text, tone, lang, y, torch.full((B,), TX), torch.full((B,), L)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    model3 = import_reference(args.ref)
    cfg = json.load(open(os.path.join(args.ref, "config.json")))
    torch.manual_seed(0)
    m = model3.NaturalSpeech2(cfg).eval()
    v = m.vits
    # every weight synthetic, the seeds of tools/make_golden_prompt.py (tests/test_prompt_cpu.py builds the mirrors from them)
    shapes = {k: tuple(t.shape) for k, t in v.state_dict().items() if k.split(".")[0] in ("ref_enc", "dp", "o_proj", "enc_p")}
    v.load_state_dict({k: torch.from_numpy(t) for k, t in synth.make_state_dict(shapes, seed=1234).items()}, strict=False)
    dshapes = {k: tuple(t.shape) for k, t in m.diff_model.state_dict().items()}
    dsd = synth.make_state_dict(dshapes, seed=1234)
    m.diff_model.load_state_dict({k: torch.from_numpy(t) for k, t in dsd.items()})
    text, tone, lang, y, x_lengths, y_lengths = inputs(v.enc_p.emb.weight.shape[0], v.enc_p.tone_emb.weight.shape[0],
                                                       v.enc_p.language_emb.weight.shape[0])
    real_like = torch.randn_like
    torch.randn_like = lambda t, **k: torch.from_numpy(synth.normal(1234, "cfg5.prior_noise", tuple(t.shape))).to(t.dtype)
    try:
        content, refer = v.infer(text, x_lengths, y, y_lengths, tone, lang)
    finally:
        torch.randn_like = real_like
    T = content.shape[2]
    x_T = torch.from_numpy(synth.normal(1234, "cfg5.x_T", (B, m.dim, T)))
    from sampler.dpm_solver import DPM_Solver, NoiseScheduleVP, model_wrapper
    ns = NoiseScheduleVP(schedule="discrete", betas=m.betas)
    model_fn = model_wrapper(m.sample_fun, ns, model_type="x_start", model_kwargs={"data": (content, refer, x_lengths, y_lengths)})
    mel = DPM_Solver(model_fn, ns, algorithm_type="dpmsolver++").sample(x_T, steps=STEPS, order=2, skip_type="time_uniform",
                                                                          method="multistep")
    # the oracle's restatement of the same chain (oracle/: prior_ref + text_enc_ref + prompt_ref + sampler_ref)
    from oracle import prior_ref, prompt_ref, sampler_ref, text_enc_ref, unet_ref
    tsd = {k: torch.from_numpy(t) for k, t in synth.make_state_dict(shapes, seed=1234).items()}
    g_ref = prior_ref.ref_enc(tsd, y).unsqueeze(-1)
    ex, em, el, emask = text_enc_ref.text_encoder(tsd, text, x_lengths, tone, lang, g_ref, cfg["vits"]["n_heads"], cfg["vits"]["n_layers"],
                                                  cfg["vits"]["kernel_size"])
    zo, _, ylen_o, _ = prior_ref.infer_from_encoder(tsd, ex, em, el, emask, x_lengths, y, y_lengths,
                                                    lambda shp: torch.from_numpy(synth.normal(1234, "cfg5.prior_noise", shp)))
    dcfg = cfg["diffusion_encoder"]
    H = dcfg["hidden_channels"]
    ucfg = unet_ref.default_config(dcfg["in_channels"] + H, dcfg["out_channels"], (128, 256, 384, 512), H, dcfg["n_heads"], 8, 2, 64)
    dtsd = {k: torch.from_numpy(t) for k, t in dsd.items()}
    mel_o = sampler_ref.dpm_solver_pp_sample(lambda xx, t_in: prompt_ref.diffusion_encoder_forward(dtsd, ucfg, xx, zo, y, y_lengths, t_in),
                                             m.betas, x_T, STEPS, 2, "time_uniform")
    frames = torch.clamp_min(torch.sum(torch.ceil(torch.exp(v.dp(ex, x_lengths, y, y_lengths)) * emask), [1, 2]), 1).long()
    print("config 5 (B=%d, Tx=%d, L=%d): frames per utterance %s -> padded T = %d ; |mel| %.3f" % (B, TX, L, frames.tolist(), T, float(mel.abs().mean())))
    print("oracle vs reference: content %.2e  mel %.2e" % (rel(zo.numpy(), content.numpy()), rel(mel_o.numpy(), mel.numpy())))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config5_b16.npz"), B=B, TX=TX, L=L, steps=STEPS, T=T,
                        text=text.numpy(), tone=tone.numpy(), language=lang.numpy(), frames=frames.numpy(), mel=mel.numpy(),
                        diffusion_encoder=np.array(repr(dcfg)), timesteps=cfg["train"]["timesteps"])


if __name__ == "__main__":
    main()
