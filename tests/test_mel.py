"""CPU: mel front-end - known-answer properties only; parity with torchaudio is UNPINNED
(torchaudio is absent here and on the GPU box, see diff_vits_amd/mel.py)."""
import math

import numpy as np
import torch

from diff_vits_amd.mel import MelSpectrogram, melscale_fbanks, reference_mel_prompt


def test_filterbank_shape_and_partition():
    fb = melscale_fbanks(513, 0.0, 12000.0, 100, 24000)
    assert tuple(fb.shape) == (513, 100) and float(fb.min()) >= 0.0
    centre = fb.argmax(dim=0)
    assert torch.all(centre[1:] >= centre[:-1])                       # filters ordered by frequency
    # triangular filters on the HTK grid overlap so that interior bins sum to 1 (no normalisation)
    s = fb.sum(dim=1)
    assert float((s[20:480] - 1.0).abs().max()) < 1e-4


def test_pure_tone_peaks_in_the_right_filter():
    sr, f0 = 24000, 3000.0
    t = torch.arange(sr, dtype=torch.float32) / sr
    mel = MelSpectrogram()(torch.sin(2 * math.pi * f0 * t).unsqueeze(0))
    assert tuple(mel.shape) == (1, 100, sr // 256 + 1)                # center=True: 1 + n // hop frames
    fb = melscale_fbanks(513, 0.0, 12000.0, 100, sr)
    want = int(fb[int(round(f0 / (sr / 1024))), :].argmax())
    assert int(mel[0, :, 40].argmax()) == want
    # Hann window, unit-amplitude tone: the magnitude peak is ~ n_fft / 4 and the mel energy of the frame close to it
    assert 200.0 < float(mel[0, :, 40].sum()) < 520.0


def test_reference_mel_prompt_layout():
    x = torch.from_numpy(np.random.RandomState(0).randn(2, 5000).astype(np.float32))
    r = reference_mel_prompt(x)
    assert tuple(r.shape) == (2, 100, 5000 // 256 + 1) and torch.isfinite(r).all() and float(r.min()) >= math.log(1e-7) - 1e-6
