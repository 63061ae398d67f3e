"""Host mirror of the reference's inference script glue (tts_infer.py:46-81): build the
model from a `{'step', 'model'}` checkpoint (model3.py:1326-1345), turn the reference audio into the log-mel prompt and
call `model.sample(phoneme, refer, phoneme_length, refer_length, tone, language, vocos)`.

What stays the reference's own: the text front-end (`text/`: G2P, cleaners, the symbol table) and the vocoder (Vocos) -
both are outside the diffusion hot path and their third-party dependencies are absent from this image.  So:
  * the size of the symbol table is taken from the checkpoint (`vits.enc_p.emb.weight`) unless `n_vocab` is passed;
  * `synthesize` takes batches whose reference audio is either a waveform tensor at 24 kHz (mel front-end of mel.py,
    parity unpinned there) or a ready log-mel prompt `[1, 100, L]`; a PATH needs torchaudio for decoding / resampling,
    exactly as in the reference, and raises ImportError when it is missing.
Training-only checkpoint entries (the posterior encoder `vits.enc_q.*`) are not
part of the inference model and are skipped when loading; every key the inference model owns must be present.
"""
import torch

from .mel import reference_mel_prompt
from .model3 import VITS, NaturalSpeech2

# reference state-dict entries that only training uses: the posterior encoder (model3.py:704-712; infer never calls it)
TRAINING_ONLY_PREFIXES = ("vits.enc_q.",)


def build_model(cfg, n_vocab, backend=None):
    """NaturalSpeech2(cfg) as model3.py:955-975 builds it: VITS(len(symbols), window_size // 2 + 1, **cfg['vits'])."""
    vits = VITS(n_vocab, cfg["data"]["window_size"] // 2 + 1, backend=backend, **cfg["vits"])
    return NaturalSpeech2(cfg, vits=vits, backend=backend)


def load_state(model, state_dict):
    """load_state_dict that tolerates the training-only entries and nothing else.  Returns the skipped keys."""
    own = model.state_dict()
    skipped = [k for k in state_dict if k not in own]
    bad = [k for k in skipped if not k.startswith(TRAINING_ONLY_PREFIXES)]
    if bad:
        raise RuntimeError("checkpoint entries unknown to the inference model: %s" % ", ".join(sorted(bad)[:8]))
    missing = [k for k in own if k not in state_dict]
    if missing:
        raise RuntimeError("checkpoint lacks %d entries of the inference model, e.g. %s" % (len(missing), ", ".join(missing[:8])))
    model.load_state_dict({k: v for k, v in state_dict.items() if k in own})
    return skipped


def load_model(model_path, device, cfg, n_vocab=None, backend=None):
    """tts_infer.py:76-81: torch.load -> NaturalSpeech2(cfg) -> load_state_dict(data['model']) -> .to(device).eval()."""
    data = torch.load(model_path, map_location="cpu", weights_only=True)
    if not isinstance(data, dict) or "model" not in data:
        raise ValueError("%s is not a {'step', 'model'} checkpoint (model3.py:1326-1333)" % (model_path,))
    sd = data["model"]
    if n_vocab is None:
        if "vits.enc_p.emb.weight" not in sd:
            raise ValueError("checkpoint has no vits.enc_p.emb.weight; pass n_vocab")
        n_vocab = sd["vits.enc_p.emb.weight"].shape[0]
    model = build_model(cfg, n_vocab, backend=backend)
    load_state(model, sd)
    return model.to(device).eval()


def save_checkpoint(model, step, path):
    """The reference trainer's format (model3.py:1326-1333): {'step': int, 'model': state_dict}."""
    torch.save({"step": int(step), "model": model.state_dict()}, str(path))


def refer_prompt(refer, device):
    """Reference audio -> log-mel prompt [1, 100, L] (tts_infer.py:54-67).  `refer`: path, waveform [1, n] at 24 kHz,
    or an already computed log-mel [1, 100, L]."""
    if isinstance(refer, (str, bytes)) or hasattr(refer, "__fspath__"):
        try:
            import torchaudio
            import torchaudio.transforms as T
        except ImportError as e:
            raise ImportError("decoding a reference audio file needs torchaudio (as tts_infer.py:54-55); pass a 24 kHz "
                              "waveform tensor or a log-mel prompt instead") from e
        audio, sr = torchaudio.load(refer)
        refer = T.Resample(sr, 24000)(audio)
    refer = torch.as_tensor(refer)
    if refer.dim() == 3:
        return refer.to(device=device, dtype=torch.float32)
    if refer.dim() != 2:
        raise ValueError("reference audio must be [channels, samples] or a log-mel [1, 100, L], got %s" % (tuple(refer.shape),))
    return reference_mel_prompt(refer.to(device=device, dtype=torch.float32))


def synthesize(model, cfg, vocos, batchs, control_values=None, device="cuda", prompt_length="reference", **sample_kw):
    """tts_infer.py:46-75 with the same batch tuples `(phoneme, tone, language, refer, phoneme_length)`; returns the last
    batch's samples like the reference (and its mel as a second value).  `control_values` is accepted and, as in the
    reference, unused.  Extra keywords (`sample_method`, `noise`, `prior_noise`) go to `model.sample`.

    prompt_length: tts_infer.py:68 passes `refer.size(1)` - the mel CHANNEL count, 100 - as the prompt length, so the
    prompt masks cover the first 100 frames whatever the audio's length.  "reference" keeps that; "frames" passes the
    real frame count `refer.size(2)`."""
    if prompt_length not in ("reference", "frames"):
        raise ValueError("prompt_length must be 'reference' or 'frames'")
    samples = mel = None
    for phoneme, tone, language, refer, phoneme_length in batchs:
        phoneme, tone, language = phoneme.to(device), tone.to(device), language.to(device)
        phoneme_length = torch.as_tensor(phoneme_length, dtype=torch.long).to(device)
        spec = refer_prompt(refer, device)
        refer_length = torch.tensor([spec.size(1) if prompt_length == "reference" else spec.size(2)]).to(device)
        with torch.no_grad():
            samples, mel = model.sample(phoneme, spec, phoneme_length, refer_length, tone, language, vocos, **sample_kw)
        samples = samples.detach().cpu()
    return samples, mel
