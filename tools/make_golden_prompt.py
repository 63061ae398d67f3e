#!/usr/bin/env python3
"""Generate tests/golden/prompt_*.npz by stub-IMPORTING the reference's model3.Diffusion_Encoder (build container only).

model3.py imports packages that are absent here (vocos, torchaudio, ema_pytorch, numba, librosa, tensorboard): they
are replaced by MagicMock in sys.modules before the import; none of them is touched by
Diffusion_Encoder / PromptEncoder.  Only seeds, shapes and small outputs are stored.
Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_prompt.py [--ref /root/reference]
"""
import argparse
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import diff_vits_amd  # noqa: E402,F401
from diff_vits_amd import synth  # noqa: E402
from oracle import prompt_ref, unet_ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# name: (Diffusion_Encoder kwargs, B, T, L, prompt lengths, timesteps)
CASES = {
    "cfg": (dict(in_channels=100, out_channels=100, hidden_channels=128, n_heads=8, p_dropout=0.2), 2, 64, 40, [40, 27],
            [949.05, 911.55]),
    "long": (dict(in_channels=100, out_channels=100, hidden_channels=128, n_heads=8, p_dropout=0.2), 3, 32, 75, [75, 1, 50],
             [500.0, 20.25, 999.0]),
}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def import_reference(ref):
    import accelerate  # noqa: F401  (real, must be imported before the stubs)
    for m in ("vocos", "torchaudio", "torchaudio.transforms", "ema_pytorch", "numba", "librosa",
              "torch.utils.tensorboard", "matplotlib", "matplotlib.pyplot", "monotonic_align", "monotonic_align.core"):
        if m not in sys.modules:
            try:
                __import__(m)
            except Exception:
                sys.modules[m] = MagicMock()
    sys.modules["numba"].jit = lambda *a, **k: (lambda f: f)
    sys.path.insert(0, ref)
    import model3
    return model3


def inputs(kw, B, T, L, seed=1234):
    x = synth.normal(seed, "pe.x", (B, kw["in_channels"], T))
    cond = synth.normal(seed, "pe.cond", (B, kw["hidden_channels"], T))
    prompt = synth.normal(seed, "pe.prompt", (B, 100, L))
    return x, cond, prompt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    model3 = import_reference(args.ref)
    os.makedirs(GOLD, exist_ok=True)
    for name, (kw, B, T, L, lens, ts) in CASES.items():
        m = model3.Diffusion_Encoder(**kw).eval()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = synth.make_state_dict(shapes, seed=1234)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        x, cond, prompt = inputs(kw, B, T, L)
        lengths = torch.tensor(lens, dtype=torch.int64)
        t = torch.tensor(ts, dtype=torch.float32)
        tx, tc, tp = torch.from_numpy(x), torch.from_numpy(cond), torch.from_numpy(prompt)
        enc_ref = m.prompt_encoder(tp, lengths)                       # [B, H, L]
        y_ref = m(tx, (tc, tp, None, lengths), t)                     # [B, C, T]
        # oracle against the reference
        tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
        pe = {k[len("prompt_encoder."):]: v for k, v in tsd.items() if k.startswith("prompt_encoder.")}
        probes = {}
        enc_or = prompt_ref.prompt_encoder(pe, tp, lengths, probes=probes)
        H = kw["hidden_channels"]
        ucfg = unet_ref.default_config(kw["in_channels"] + H, kw["out_channels"], (128, 256, 384, 512), H, kw["n_heads"], 8, 2, 64)
        y_or = prompt_ref.diffusion_encoder_forward(tsd, ucfg, tx, tc, tp, lengths, t)
        print("%-6s prompt-encoder oracle vs reference %.2e   diffusion-encoder oracle vs reference %.2e   (|enc| %.3f)"
              % (name, rel(enc_or.numpy(), enc_ref.numpy()), rel(y_or.numpy(), y_ref.numpy()), float(enc_ref.abs().mean())))
        n_pe = sum(int(np.prod(s)) for k, s in shapes.items() if k.startswith("prompt_encoder."))
        np.savez_compressed(
            os.path.join(GOLD, "prompt_%s.npz" % name),
            kwargs=np.array(repr(kw)), B=B, T=T, L=L, lengths=np.array(lens, np.int64), t=np.array(ts, np.float32),
            enc=enc_ref.numpy(), y=y_ref.numpy(), n_params_prompt_encoder=n_pe,
            names=np.array(sorted(k for k in shapes if k.startswith("prompt_encoder."))),
            shapes=np.array([repr(shapes[k]) for k in sorted(shapes) if k.startswith("prompt_encoder.")]),
            **{"probe_" + k: v.numpy() for k, v in probes.items()})
    sample_golden(model3, args.ref)
    prior_golden(model3, args.ref)


def sample_golden(model3, ref):
    """NaturalSpeech2.sample (model3.py:1118-1203), 'unipc' branch, with the prior replaced by fixed synthetic
    (content, refer), torch.randn by a fixed noise tensor and vocos by a pass-through: pins the orchestration row
    (schedule buffers -> NoiseScheduleVP -> model_wrapper(sample_fun) -> UniPC bh2, 30 steps, order 2)."""
    import json
    from oracle import sample_ref
    cfg = json.load(open(os.path.join(ref, "config.json")))
    B, T, L = 1, 48, 30
    m = model3.NaturalSpeech2(cfg).eval()
    dshapes = {k: tuple(v.shape) for k, v in m.diff_model.state_dict().items()}
    dsd = synth.make_state_dict(dshapes, seed=1234)
    m.diff_model.load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()})
    content = torch.from_numpy(synth.normal(1234, "ns2.content", (B, cfg["diffusion_encoder"]["hidden_channels"], T)))
    refer = torch.from_numpy(synth.normal(1234, "ns2.refer", (B, 100, L)))
    noise = torch.from_numpy(synth.normal(1234, "ns2.noise", (B, cfg["diffusion_encoder"]["in_channels"], T)))
    text_lengths, spec_lengths = torch.tensor([17]), torch.tensor([L - 4])
    m.vits.infer = lambda *a, **k: (content, refer)

    class PassThroughVocoder:
        def to(self, device):
            return self

        def decode(self, mel):
            return mel.mean(dim=1, keepdim=True)

    real_randn = torch.randn
    torch.randn = lambda *a, **k: noise.clone() if tuple(a[0] if isinstance(a[0], (tuple, list)) else a) == tuple(noise.shape) \
        else real_randn(*a, **k)
    try:
        audio, mel = m.sample(None, refer, text_lengths, spec_lengths, None, None, PassThroughVocoder(), sample_method="unipc")
    finally:
        torch.randn = real_randn
    # oracle restatement of the same orchestration
    tsd = {k: torch.from_numpy(v) for k, v in dsd.items()}
    mel_or = sample_ref.sample_mel(tsd, cfg["diffusion_encoder"], content, refer, text_lengths, spec_lengths, noise,
                                   sample_method="unipc", timesteps=cfg["train"]["timesteps"])
    bufs_or = sample_ref.schedule_buffers(cfg["train"]["timesteps"])
    bufs = {k: v.numpy() for k, v in m.state_dict().items() if k in bufs_or}
    assert len(bufs) == len(bufs_or)
    worst = max(rel(bufs_or[k].numpy(), bufs[k]) for k in bufs)
    print("sample  unipc mel: oracle vs reference %.2e ; schedule buffers (%d) worst %.2e ; |mel| %.3f"
          % (rel(mel_or.numpy(), mel.numpy()), len(bufs), worst, float(mel.abs().mean())))
    np.savez_compressed(os.path.join(GOLD, "sample_unipc.npz"), B=B, T=T, L=L, text_lengths=text_lengths.numpy(),
                        spec_lengths=spec_lengths.numpy(), mel=mel.numpy(), audio=audio.numpy(),
                        diffusion_encoder=np.array(repr(cfg["diffusion_encoder"])), timesteps=cfg["train"]["timesteps"],
                        **{"buf_" + k: v for k, v in bufs.items()})


def prior_golden(model3, ref):
    """VITS.infer (model3.py:817-860) with synthetic weights in ref_enc / dp / o_proj, the reference's own randomly
    initialised text encoder (its outputs are captured and stored: the tests start from them), torch.randn_like
    replaced by the seeded generator."""
    import json
    from oracle import prior_ref
    cfg = json.load(open(os.path.join(ref, "config.json")))
    torch.manual_seed(0)
    m = model3.NaturalSpeech2(cfg).eval()
    v = m.vits
    shapes = {k: tuple(t.shape) for k, t in v.state_dict().items() if k.split(".")[0] in ("ref_enc", "dp", "o_proj", "enc_p")}
    sd = synth.make_state_dict(shapes, seed=1234)
    missing = v.load_state_dict({k: torch.from_numpy(t) for k, t in sd.items()}, strict=False)
    assert not [k for k in missing.unexpected_keys]
    B, Tx, L = 2, 11, 36
    n_sym = v.enc_p.emb.weight.shape[0]
    text = torch.from_numpy((synth.uniform(1234, "prior.text", (B, Tx)) * 0.5 + 0.5) * (n_sym - 1)).long()
    tone = torch.from_numpy((synth.uniform(1234, "prior.tone", (B, Tx)) * 0.5 + 0.5) * (v.enc_p.tone_emb.weight.shape[0] - 1)).long()
    lang = torch.from_numpy((synth.uniform(1234, "prior.lang", (B, Tx)) * 0.5 + 0.5) * (v.enc_p.language_emb.weight.shape[0] - 1)).long()
    x_lengths = torch.tensor([Tx, Tx - 3])
    y = torch.from_numpy(synth.normal(1234, "prior.refer", (B, 100, L)))
    y_lengths = torch.tensor([L, L - 9])
    captured = {}
    enc_forward = v.enc_p.forward

    def enc_hook(*a, **k):
        out = enc_forward(*a, **k)
        captured["enc"] = [t.clone() for t in out]
        return out
    v.enc_p.forward = enc_hook
    real = torch.randn_like
    torch.randn_like = lambda t, **k: torch.from_numpy(synth.normal(1234, "prior.noise", tuple(t.shape))).to(t.dtype)
    try:
        z, yy = v.infer(text, x_lengths, y, y_lengths, tone, lang)
    finally:
        torch.randn_like = real
        v.enc_p.forward = enc_forward
    x, m_p, logs_p, x_mask = captured["enc"]
    tsd = {k: torch.from_numpy(t) for k, t in sd.items()}
    from oracle import text_enc_ref
    g_ref = prior_ref.ref_enc(tsd, y).unsqueeze(-1)
    eo = text_enc_ref.text_encoder(tsd, text, x_lengths, tone, lang, g_ref, cfg["vits"]["n_heads"], cfg["vits"]["n_layers"],
                                   cfg["vits"]["kernel_size"])
    print("prior  text encoder: oracle vs reference x %.2e  m %.2e  logs %.2e" % (
        rel(eo[0].numpy(), x.numpy()), rel(eo[1].numpy(), m_p.numpy()), rel(eo[2].numpy(), logs_p.numpy())))
    zo, _, ylen_o, logw_o = prior_ref.infer_from_encoder(
        tsd, x, m_p, logs_p, x_mask, x_lengths, y, y_lengths,
        lambda shp: torch.from_numpy(synth.normal(1234, "prior.noise", shp)))
    print("prior  z: oracle vs reference %.2e ; frames %s ; |z| %.3f" % (rel(zo.numpy(), z.numpy()), list(ylen_o.numpy()), float(z.abs().mean())))
    # ---- the whole tts_infer.py call, phoneme ids -> mel: NaturalSpeech2.sample('unipc') with the real vits.infer (B = 1:
    #      the reference's UniPC wrapper only broadcasts there), every weight synthetic, all noise from the seeded generator
    dshapes = {k: tuple(t.shape) for k, t in m.diff_model.state_dict().items()}
    m.diff_model.load_state_dict({k: torch.from_numpy(t) for k, t in synth.make_state_dict(dshapes, seed=1234).items()})

    class PassThroughVocoder:
        def to(self, device):
            return self

        def decode(self, mel):
            return mel.mean(dim=1, keepdim=True)
    real_like, real_randn = torch.randn_like, torch.randn
    torch.randn_like = lambda t, **k: torch.from_numpy(synth.normal(1234, "full.prior_noise", tuple(t.shape))).to(t.dtype)
    torch.randn = lambda *a, **k: torch.from_numpy(synth.normal(1234, "full.x_T", tuple(a[0]) if isinstance(a[0], (tuple, list)) else tuple(a)))
    try:
        audio_f, mel_f = m.sample(text[:1], y[:1], x_lengths[:1], y_lengths[:1], tone[:1], lang[:1], PassThroughVocoder(), sample_method="unipc")
    finally:
        torch.randn_like, torch.randn = real_like, real_randn
    print("full   ids -> mel: frames %d ; |mel| %.3f" % (mel_f.shape[2], float(mel_f.abs().mean())))
    np.savez_compressed(os.path.join(GOLD, "sample_full.npz"), mel=mel_f.numpy(), audio=audio_f.numpy(),
                        diffusion_encoder=np.array(repr(cfg["diffusion_encoder"])), timesteps=cfg["train"]["timesteps"])
    np.savez_compressed(os.path.join(GOLD, "prior_infer.npz"), enc_x=x.numpy(), enc_m_p=m_p.numpy(), enc_logs_p=logs_p.numpy(),
                        enc_x_mask=x_mask.numpy(), x_lengths=x_lengths.numpy(), y_lengths=y_lengths.numpy(), L=L,
                        z=z.numpy(), y_len_out=ylen_o.numpy(), logw=logw_o.numpy(), text=text.numpy(), tone=tone.numpy(),
                        language=lang.numpy(), n_vocab=n_sym, n_tones=v.enc_p.tone_emb.weight.shape[0],
                        n_languages=v.enc_p.language_emb.weight.shape[0], vits_kwargs=np.array(repr(cfg["vits"])),
                        names=np.array(sorted(shapes)), shapes=np.array([repr(shapes[k]) for k in sorted(shapes)]),
                        vits_cfg=np.array(repr({k: cfg["vits"][k] for k in ("inter_channels", "hidden_channels")})))


if __name__ == "__main__":
    main()
