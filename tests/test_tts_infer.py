"""Host mirror of the reference's inference-script glue (diff_vits_amd/tts_infer.py; reference tts_infer.py:46-81 and the
trainer's checkpoint format model3.py:1326-1345): a `{'step', 'model'}` checkpoint with the reference's parameter names is
loaded into the mirrors and `synthesize` reproduces the reference's own `sample()` output (tests/golden/sample_full.npz)."""
import ast

import numpy as np
import pytest
import torch

from diff_vits_amd import synth, tts_infer
from test_prompt_cpu import PassThroughVocoder, diffusion_state_dict, prior_case, rel_l2


def chain_cfg(g, gf):
    vits = dict(ast.literal_eval(str(g["vits_kwargs"])), n_tones=int(g["n_tones"]), n_languages=int(g["n_languages"]))
    return {"data": {"window_size": 1024}, "vits": vits, "diffusion_encoder": ast.literal_eval(str(gf["diffusion_encoder"])),
            "train": {"timesteps": int(gf["timesteps"])}}


def reference_checkpoint(gold, tmp_path, extra=None):
    """A checkpoint as the reference trainer writes it: every inference weight (synthetic, the fixtures' seeds) under the
    reference's names, the schedule buffers, and a training-only posterior-encoder entry."""
    g, sd, y = prior_case(gold)
    gf = gold("sample_full.npz")
    cfg = chain_cfg(g, gf)
    model = tts_infer.build_model(cfg, int(g["n_vocab"]), backend="torch")
    state = {k: v for k, v in model.state_dict().items() if not k.startswith(("vits.", "diff_model."))}   # buffers
    state.update({"vits." + k: torch.from_numpy(v) for k, v in sd.items()})
    state.update({"diff_model." + k: torch.from_numpy(v) for k, v in diffusion_state_dict(cfg["diffusion_encoder"]).items()})
    state["vits.enc_q.pre.weight"] = torch.zeros(4, 4, 1)
    state.update(extra or {})
    path = tmp_path / "model-7.pt"
    torch.save({"step": 7, "model": state}, str(path))
    return path, cfg, g, gf, y


def test_load_model_and_synthesize_match_reference_sample(gold, tmp_path):
    path, cfg, g, gf, y = reference_checkpoint(gold, tmp_path)
    model = tts_infer.load_model(path, "cpu", cfg, backend="torch")            # n_vocab from the checkpoint's embedding
    assert not model.training and model.vits.enc_p.emb.weight.shape[0] == int(g["n_vocab"])
    T = gf["mel"].shape[2]
    x_T = torch.from_numpy(synth.normal(1234, "full.x_T", (1, cfg["diffusion_encoder"]["in_channels"], T)))
    pn = torch.from_numpy(synth.normal(1234, "full.prior_noise", (1, 128, T)))
    batch = (torch.from_numpy(g["text"][:1]), torch.from_numpy(g["tone"][:1]), torch.from_numpy(g["language"][:1]),
             torch.from_numpy(y[:1]), g["x_lengths"][:1].tolist())
    audio, mel = tts_infer.synthesize(model, cfg, PassThroughVocoder(), [batch], None, "cpu", prompt_length="frames",
                                      sample_method="unipc", noise=x_T, prior_noise=pn)
    assert rel_l2(mel.numpy(), gf["mel"]) < 2e-5 and rel_l2(audio.numpy(), gf["audio"]) < 2e-5
    # the reference script's own prompt length (refer.size(1) = 100 mel channels) covers this 36-frame prompt entirely
    audio2, _ = tts_infer.synthesize(model, cfg, PassThroughVocoder(), [batch], None, "cpu", sample_method="unipc",
                                     noise=x_T, prior_noise=pn)
    assert torch.equal(audio2, audio)


def test_checkpoint_round_trip_and_errors(gold, tmp_path):
    path, cfg, g, gf, y = reference_checkpoint(gold, tmp_path)
    model = tts_infer.load_model(path, "cpu", cfg, backend="torch")
    tts_infer.save_checkpoint(model, 8, tmp_path / "again.pt")
    data = torch.load(str(tmp_path / "again.pt"), weights_only=True)
    assert data["step"] == 8 and sorted(data["model"]) == sorted(model.state_dict())
    again = tts_infer.load_model(tmp_path / "again.pt", "cpu", cfg, backend="torch")
    assert all(torch.equal(v, again.state_dict()[k]) for k, v in model.state_dict().items())

    torch.save({"weights": {}}, str(tmp_path / "bad.pt"))
    with pytest.raises(ValueError):
        tts_infer.load_model(tmp_path / "bad.pt", "cpu", cfg)
    sd = dict(data["model"])
    sd.pop("diff_model.unet.conv_in.weight")
    with pytest.raises(RuntimeError, match="lacks"):
        tts_infer.load_state(model, sd)
    with pytest.raises(RuntimeError, match="unknown"):
        tts_infer.load_state(model, dict(data["model"], **{"vits.flow.x": torch.zeros(1)}))
    with pytest.raises(ImportError):
        tts_infer.refer_prompt("prompt.wav", "cpu")                             # torchaudio is not in this image
    with pytest.raises(ValueError):
        tts_infer.synthesize(model, cfg, None, [], prompt_length="samples")


def test_waveform_prompt_goes_through_the_mel_front_end():
    wave = torch.from_numpy(synth.normal(3, "wave", (1, 24000 // 4))) * 0.1
    spec = tts_infer.refer_prompt(wave, "cpu")
    assert spec.shape == (1, 100, 24000 // 4 // 256 + 1) and spec.dtype == torch.float32
    assert float(spec.min()) >= float(np.log(1e-7)) - 1e-4
    assert tts_infer.refer_prompt(spec, "cpu") is spec or torch.equal(tts_infer.refer_prompt(spec, "cpu"), spec)
