"""CPU: the prompt encoder / Diffusion_Encoder row — oracle against the goldens captured from
the stub-imported reference (tools/make_golden_prompt.py), mirror module layout, and the mirror's explicit torch
backend against the same goldens."""
import ast

import numpy as np
import pytest
import torch

from conftest import rel_l2
from diff_vits_amd import synth
from diff_vits_amd.model3 import Diffusion_Encoder, PromptEncoder
from oracle import prompt_ref, unet_ref

CASES = ("cfg", "long")


def prompt_case(gold, name):
    g = gold("prompt_%s.npz" % name)
    kw = ast.literal_eval(str(g["kwargs"]))
    B, T, L = int(g["B"]), int(g["T"]), int(g["L"])
    x = synth.normal(1234, "pe.x", (B, kw["in_channels"], T))
    cond = synth.normal(1234, "pe.cond", (B, kw["hidden_channels"], T))
    prompt = synth.normal(1234, "pe.prompt", (B, 100, L))
    return g, kw, x, cond, prompt, g["lengths"], g["t"]


def diffusion_state_dict(kw):
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in Diffusion_Encoder(backend="torch", **kw).state_dict().items()}
    return synth.make_state_dict(shapes, seed=1234)


@pytest.mark.parametrize("name", CASES)
def test_oracle_prompt_encoder_matches_reference(gold, name):
    g, kw, x, cond, prompt, lengths, t = prompt_case(gold, name)
    sd = {k[len("prompt_encoder."):]: torch.from_numpy(v) for k, v in diffusion_state_dict(kw).items()
          if k.startswith("prompt_encoder.")}
    probes = {}
    with torch.no_grad():
        enc = prompt_ref.prompt_encoder(sd, torch.from_numpy(prompt), torch.from_numpy(lengths), probes=probes)
    assert rel_l2(enc.numpy(), g["enc"]) < 1e-6
    for k, v in probes.items():
        assert rel_l2(v.numpy(), g["probe_" + k]) < 1e-6
    # padding frames are exactly zero, valid frames are not
    for b, n in enumerate(lengths):
        assert np.all(enc.numpy()[b, :, int(n):] == 0) and np.any(enc.numpy()[b, :, :int(n)] != 0)


def test_oracle_diffusion_encoder_matches_reference(gold):
    g, kw, x, cond, prompt, lengths, t = prompt_case(gold, "cfg")
    sd = {k: torch.from_numpy(v) for k, v in diffusion_state_dict(kw).items()}
    H = kw["hidden_channels"]
    ucfg = unet_ref.default_config(kw["in_channels"] + H, kw["out_channels"], (128, 256, 384, 512), H, kw["n_heads"], 8, 2, 64)
    with torch.no_grad():
        y = prompt_ref.diffusion_encoder_forward(sd, ucfg, torch.from_numpy(x), torch.from_numpy(cond), torch.from_numpy(prompt),
                                                 torch.from_numpy(lengths), torch.from_numpy(t))
    assert rel_l2(y.numpy(), g["y"]) < 1e-6


def test_mirror_layout_matches_reference(gold):
    g = gold("prompt_cfg.npz")
    with torch.device("meta"):
        m = PromptEncoder(100, 128, 128, 4, 0.2, backend="torch")
    sd = m.state_dict()
    names = [str(n)[len("prompt_encoder."):] for n in g["names"]]
    assert sorted(sd) == sorted(names)
    for n, s in zip(names, g["shapes"]):
        assert tuple(sd[n].shape) == ast.literal_eval(str(s)), n
    assert sum(v.numel() for v in sd.values()) == int(g["n_params_prompt_encoder"]) == 2918344 - 0
    with torch.device("meta"):
        d = Diffusion_Encoder(in_channels=100, out_channels=100, hidden_channels=128, n_heads=8, p_dropout=0.2, backend="torch")
    assert len([k for k in d.state_dict() if k.startswith("unet.")]) == 701
    with pytest.raises(ValueError):
        PromptEncoder(100, 128, 128, 4, backend="cuda")


@pytest.mark.parametrize("name", CASES)
def test_mirror_torch_backend_matches_reference(gold, name):
    g, kw, x, cond, prompt, lengths, t = prompt_case(gold, name)
    m = Diffusion_Encoder(backend="torch", **kw).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in diffusion_state_dict(kw).items()})
    with torch.no_grad():
        enc = m.prompt_encoder(torch.from_numpy(prompt), torch.from_numpy(lengths))
        assert rel_l2(enc.numpy(), g["enc"]) < 1e-6
        if name == "cfg":
            data = (torch.from_numpy(cond), torch.from_numpy(prompt), None, torch.from_numpy(lengths))
            y = m(torch.from_numpy(x), data, torch.from_numpy(t))
            assert rel_l2(y.numpy(), g["y"]) < 1e-6
            y2 = m(torch.from_numpy(x), data, torch.from_numpy(t))       # cached conditioning: identical
            assert torch.equal(y, y2)


def test_conditioning_cache_survives_address_reuse(gold):
    """Regression (round-1 advisor finding): the step-invariant conditioning is cached per (prompt, lengths) pair.  A key
    of data_ptr()/_version alone is met by a NEW same-shape prompt allocated at the freed address of the previous one
    (the allocator recycles blocks), which then silently got the previous speaker's encoder states.  The cache entry
    now owns the tensors: freed-and-reallocated prompts always re-encode; in-place edits (version bump) too."""
    g, kw, x, cond, prompt, lengths, t = prompt_case(gold, "cfg")
    m = Diffusion_Encoder(backend="torch", **kw).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in diffusion_state_dict(kw).items()})
    ln = torch.from_numpy(lengths)
    with torch.no_grad():
        for i in range(6):
            spec = torch.from_numpy(prompt).clone() * float(i + 1)       # same shape, often the same address
            enc, _ = m._conditioning(spec, ln, torch.float32)
            want = m.prompt_encoder.encode_channels_last(spec, ln) * sequence_mask_like(ln, spec.size(2))
            assert torch.equal(enc, want), i
            del spec
        spec = torch.from_numpy(prompt).clone()
        e1 = m._conditioning(spec, ln, torch.float32)[0]
        assert m._conditioning(spec, ln, torch.float32)[0] is e1        # hit: same objects, unmodified
        spec.mul_(2.0)                                                  # in-place edit -> new version -> re-encode
        assert not torch.equal(m._conditioning(spec, ln, torch.float32)[0], e1)


def sequence_mask_like(lengths, L):
    return (torch.arange(L)[None, :] < lengths[:, None]).unsqueeze(-1).to(torch.float32)


# ---- NaturalSpeech2.sample orchestration ------------------------------------------------------
def sample_case(gold):
    from diff_vits_amd.model3 import NaturalSpeech2
    g = gold("sample_unipc.npz")
    dcfg = ast.literal_eval(str(g["diffusion_encoder"]))
    cfg = {"diffusion_encoder": dcfg, "train": {"timesteps": int(g["timesteps"])}}
    B, T, L = int(g["B"]), int(g["T"]), int(g["L"])
    content = synth.normal(1234, "ns2.content", (B, dcfg["hidden_channels"], T))
    refer = synth.normal(1234, "ns2.refer", (B, 100, L))
    noise = synth.normal(1234, "ns2.noise", (B, dcfg["in_channels"], T))
    return g, cfg, NaturalSpeech2, content, refer, noise


class PassThroughVocoder:
    def to(self, device):
        return self

    def decode(self, mel):
        return mel.mean(dim=1, keepdim=True)


def test_oracle_sample_orchestration_matches_reference(gold):
    from oracle import sample_ref
    g, cfg, _, content, refer, noise = sample_case(gold)
    bufs = sample_ref.schedule_buffers(cfg["train"]["timesteps"])
    for k, v in bufs.items():
        assert np.array_equal(v.numpy(), g["buf_" + k]), k
    sd = {k: torch.from_numpy(v) for k, v in diffusion_state_dict(cfg["diffusion_encoder"]).items()}
    mel = sample_ref.sample_mel(sd, cfg["diffusion_encoder"], torch.from_numpy(content), torch.from_numpy(refer),
                                torch.from_numpy(g["text_lengths"]), torch.from_numpy(g["spec_lengths"]),
                                torch.from_numpy(noise), "unipc", cfg["train"]["timesteps"])
    assert rel_l2(mel.numpy(), g["mel"]) < 1e-6


def test_mirror_sample_torch_backend_matches_reference(gold):
    g, cfg, NaturalSpeech2, content, refer, noise = sample_case(gold)

    class Prior(torch.nn.Module):
        def infer(self, text, text_lengths, spec, spec_lengths, tone, language):
            return torch.from_numpy(content), spec
    m = NaturalSpeech2(cfg, vits=Prior(), backend="torch").eval()
    for k in ("betas", "alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef2"):
        assert np.array_equal(getattr(m, k).numpy(), g["buf_" + k]), k
    m.diff_model.load_state_dict({k: torch.from_numpy(v) for k, v in diffusion_state_dict(cfg["diffusion_encoder"]).items()})
    audio, mel = m.sample(None, torch.from_numpy(refer), torch.from_numpy(g["text_lengths"]), torch.from_numpy(g["spec_lengths"]),
                          None, None, PassThroughVocoder(), sample_method="unipc", noise=torch.from_numpy(noise))
    assert rel_l2(mel.numpy(), g["mel"]) < 1e-5 and rel_l2(audio.numpy(), g["audio"]) < 1e-5
    with pytest.raises(ValueError):
        m.sample_from_prior(torch.from_numpy(content), torch.from_numpy(refer), None, torch.from_numpy(g["spec_lengths"]),
                            sample_method="ddim")
    with pytest.raises(RuntimeError):
        NaturalSpeech2(cfg, backend="torch").sample(None, None, None, None, None, None, None)


# ---- the VITS prior from the text encoder's outputs onward ------------------------------
def prior_case(gold):
    g = gold("prior_infer.npz")
    names = [str(n) for n in g["names"]]
    shapes = {n: ast.literal_eval(str(s)) for n, s in zip(names, g["shapes"])}
    sd = synth.make_state_dict(shapes, seed=1234)
    L = int(g["L"])
    y = synth.normal(1234, "prior.refer", (2, 100, L))
    return g, sd, y


def test_oracle_prior_matches_reference(gold):
    from oracle import prior_ref
    g, sd, y = prior_case(gold)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        z, _, ylen, logw = prior_ref.infer_from_encoder(
            tsd, torch.from_numpy(g["enc_x"]), torch.from_numpy(g["enc_m_p"]), torch.from_numpy(g["enc_logs_p"]),
            torch.from_numpy(g["enc_x_mask"]), torch.from_numpy(g["x_lengths"]), torch.from_numpy(y),
            torch.from_numpy(g["y_lengths"]), lambda shp: torch.from_numpy(synth.normal(1234, "prior.noise", shp)))
    assert np.array_equal(ylen.numpy(), g["y_len_out"]) and rel_l2(z.numpy(), g["z"]) < 1e-6
    # the integer durations of this fixture are not within rounding of a boundary (the GPU test relies on it)
    w = np.exp(g["logw"]) * g["enc_x_mask"]
    assert np.abs(w - np.round(w))[g["enc_x_mask"] > 0].min() > 1e-3


def vits_mirror(g, sd, backend):
    from diff_vits_amd.model3 import VITS
    kw = ast.literal_eval(str(g["vits_kwargs"]))
    m = VITS(int(g["n_vocab"]), 513, n_tones=int(g["n_tones"]), n_languages=int(g["n_languages"]), backend=backend, **kw).eval()
    own = m.state_dict()
    assert sorted(own) == sorted(sd) and all(tuple(own[k].shape) == tuple(sd[k].shape) for k in sd)   # 949 reference names
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m


def test_oracle_text_encoder_matches_reference(gold):
    from oracle import prior_ref, text_enc_ref
    g, sd, y = prior_case(gold)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        gg = prior_ref.ref_enc(tsd, torch.from_numpy(y)).unsqueeze(-1)
        x, m_p, logs_p, x_mask = text_enc_ref.text_encoder(tsd, torch.from_numpy(g["text"]), torch.from_numpy(g["x_lengths"]),
                                                           torch.from_numpy(g["tone"]), torch.from_numpy(g["language"]), gg)
    for a, k in ((x, "enc_x"), (m_p, "enc_m_p"), (logs_p, "enc_logs_p"), (x_mask, "enc_x_mask")):
        assert rel_l2(a.numpy(), g[k]) < 1e-6, k


def test_mirror_prior_torch_backend_matches_reference(gold):
    from diff_vits_amd.model3 import VITS, generate_path
    g, sd, y = prior_case(gold)
    m = vits_mirror(g, sd, "torch")
    noise_full = torch.from_numpy(synth.normal(1234, "prior.noise", tuple(g["z"].shape)))
    zf, _ = m.infer(torch.from_numpy(g["text"]), torch.from_numpy(g["x_lengths"]), torch.from_numpy(y), torch.from_numpy(g["y_lengths"]),
                    torch.from_numpy(g["tone"]), torch.from_numpy(g["language"]), noise=noise_full)
    assert rel_l2(zf.numpy(), g["z"]) < 1e-5               # whole VITS.infer, text ids -> z
    noise = torch.from_numpy(synth.normal(1234, "prior.noise", tuple(g["z"].shape)))
    z, yy, ylen = m.infer_from_encoder(torch.from_numpy(g["enc_x"]), torch.from_numpy(g["enc_m_p"]), torch.from_numpy(g["enc_logs_p"]),
                                       torch.from_numpy(g["enc_x_mask"]), torch.from_numpy(g["x_lengths"]), torch.from_numpy(y),
                                       torch.from_numpy(g["y_lengths"]), noise=noise)
    assert np.array_equal(ylen.numpy(), g["y_len_out"]) and rel_l2(z.numpy(), g["z"]) < 1e-5

    class Enc(torch.nn.Module):          # any module with the reference's enc_p signature plugs in
        def forward(self, x, x_lengths, tone, language, g_):
            return (torch.from_numpy(gold("prior_infer.npz")["enc_x"]), torch.from_numpy(gold("prior_infer.npz")["enc_m_p"]),
                    torch.from_numpy(gold("prior_infer.npz")["enc_logs_p"]), torch.from_numpy(gold("prior_infer.npz")["enc_x_mask"]))
    m.enc_p = Enc()
    z2, _ = m.infer(None, torch.from_numpy(g["x_lengths"]), torch.from_numpy(y), torch.from_numpy(g["y_lengths"]), None, None, noise=noise)
    assert torch.equal(z2, z)
    with pytest.raises(RuntimeError):
        VITS(backend="torch").infer(None, None, torch.from_numpy(y), None, None, None)
    # alignment path: each output frame belongs to exactly one token, in order
    d = torch.tensor([[[2.0, 0.0, 3.0]]])
    p = generate_path(d, torch.ones(1, 1, 5, 3))
    assert p.sum().item() == 5 and p[0, 0, :, 0].tolist() == [1, 1, 0, 0, 0] and p[0, 0, :, 2].tolist() == [0, 0, 1, 1, 1]


def full_chain(gold, backend):
    """NaturalSpeech2(cfg, vits=VITS(...)) with every weight synthetic (the fixtures' seeds), as tts_infer.py builds it."""
    from diff_vits_amd.model3 import NaturalSpeech2
    g, sd, y = prior_case(gold)
    gf = gold("sample_full.npz")
    dcfg = ast.literal_eval(str(gf["diffusion_encoder"]))
    ns2 = NaturalSpeech2({"diffusion_encoder": dcfg, "train": {"timesteps": int(gf["timesteps"])}}, vits=vits_mirror(g, sd, backend),
                         backend=backend).eval()
    ns2.diff_model.load_state_dict({k: torch.from_numpy(v) for k, v in diffusion_state_dict(dcfg).items()})
    T = gf["mel"].shape[2]
    x_T = synth.normal(1234, "full.x_T", (1, dcfg["in_channels"], T))
    pn = synth.normal(1234, "full.prior_noise", (1, 128, T))
    args = (g["text"][:1], y[:1], g["x_lengths"][:1], g["y_lengths"][:1], g["tone"][:1], g["language"][:1])
    return ns2, gf, args, x_T, pn


def test_full_chain_ids_to_mel_torch_backend(gold):
    """tts_infer.py's model.sample(phoneme, refer, phoneme_length, refer_length, tone, language, vocos) on the mirrors:
    text encoder -> durations -> alignment -> o_proj -> prompt encoder -> 30-step UniPC over the UNet, against the
    reference's own sample() run here on the same synthetic weights and noise."""
    ns2, gf, args, x_T, pn = full_chain(gold, "torch")
    audio, mel = ns2.sample(*[torch.from_numpy(a) for a in args], PassThroughVocoder(), sample_method="unipc",
                            noise=torch.from_numpy(x_T), prior_noise=torch.from_numpy(pn))
    assert mel.shape == gf["mel"].shape and rel_l2(mel.numpy(), gf["mel"]) < 2e-5 and rel_l2(audio.numpy(), gf["audio"]) < 2e-5


# ---- BASELINE.json configuration 5 (B = 16, C = 100, prior -> prompt encoder -> 20-step DPM-Solver++) at reduced length ---------
def config5_case(gold):
    """Inputs of tests/golden/config5_b16.npz (tools/make_golden_config5.py: the stub-imported reference's own chain)."""
    g5 = gold("config5_b16.npz")
    B, L, T = int(g5["B"]), int(g5["L"]), int(g5["T"])
    dcfg = ast.literal_eval(str(g5["diffusion_encoder"]))
    y = synth.normal(1234, "cfg5.refer", (B, 100, L))
    x_T = synth.normal(1234, "cfg5.x_T", (B, dcfg["in_channels"], T))
    pn = synth.normal(1234, "cfg5.prior_noise", (B, 128, T))
    x_lengths = np.full((B,), int(g5["TX"]), np.int64)
    y_lengths = np.full((B,), L, np.int64)
    return g5, dcfg, y, x_T, pn, x_lengths, y_lengths


def test_oracle_config5_b16_chain_matches_reference(gold):
    """The oracle's restatement of the whole configuration-5 chain - text encoder, duration predictor, alignment, o_proj
    (oracle/text_enc_ref.py, prior_ref.py), prompt encoder inside every denoiser call (prompt_ref.py), 20-step DPM-Solver++
    (sampler_ref.py) - on sixteen utterances against the mel the imported reference produced for the same ids and seeds."""
    from oracle import prior_ref, prompt_ref, sampler_ref, text_enc_ref, unet_ref
    g5, dcfg, y, x_T, pn, x_lengths, y_lengths = config5_case(gold)
    g, sd, _ = prior_case(gold)
    kw = ast.literal_eval(str(g["vits_kwargs"]))
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    ty, tl, yl = torch.from_numpy(y), torch.from_numpy(x_lengths), torch.from_numpy(y_lengths)
    with torch.no_grad():
        g_ref = prior_ref.ref_enc(tsd, ty).unsqueeze(-1)
        ex, em, el, emask = text_enc_ref.text_encoder(tsd, torch.from_numpy(g5["text"]), tl, torch.from_numpy(g5["tone"]),
                                                      torch.from_numpy(g5["language"]), g_ref, kw["n_heads"], kw["n_layers"], kw["kernel_size"])
        z, _, ylen, _ = prior_ref.infer_from_encoder(tsd, ex, em, el, emask, tl, ty, yl, lambda shp: torch.from_numpy(pn))
        assert np.array_equal(ylen.numpy(), g5["frames"]) and z.shape[2] == int(g5["T"])
        H = dcfg["hidden_channels"]
        ucfg = unet_ref.default_config(dcfg["in_channels"] + H, dcfg["out_channels"], (128, 256, 384, 512), H, dcfg["n_heads"], 8, 2, 64)
        dsd = {k: torch.from_numpy(v) for k, v in diffusion_state_dict(dcfg).items()}
        betas = torch.from_numpy(np.asarray(sampler_ref_betas(int(g5["timesteps"]))))
        mel = sampler_ref.dpm_solver_pp_sample(lambda xx, t_in: prompt_ref.diffusion_encoder_forward(dsd, ucfg, xx, z, ty, yl, t_in),
                                               betas, torch.from_numpy(x_T), int(g5["steps"]), 2, "time_uniform")
    assert rel_l2(mel.numpy(), g5["mel"]) < 1e-6


def sampler_ref_betas(timesteps):
    from oracle import sample_ref
    return sample_ref.schedule_buffers(timesteps)["betas"].numpy()
