"""Mel front-end of the reference's inference script: `tts_infer.py:57-66` builds
`torchaudio.transforms.MelSpectrogram(sample_rate=24000, n_fft=1024, hop_length=256, n_mels=100, center=True, power=1)`
and feeds `log(clip(spec, 1e-7))` as the reference mel prompt `[B, 100, L]`.

PARITY UNPINNED: torchaudio is not installed in the build container or on the GPU box, so this restatement of its
documented defaults (Hann window, periodic; reflect padding; one-sided magnitude spectrum; HTK mel scale, no filter
normalisation, f_min 0, f_max sr/2) could not be compared with torchaudio output; tests/test_mel.py checks known-answer
properties only (filterbank partition of unity inside the band, a pure tone's peak bin, frame count).  torch.stft does
the transform; nothing here is on the diffusion hot path.
"""
import math

import torch


def _hz_to_mel(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """Triangular HTK-scale filterbank [n_freqs, n_mels] (torchaudio.functional.melscale_fbanks, norm=None)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m_pts = torch.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0).to(torch.float32)


class MelSpectrogram(torch.nn.Module):
    """`spec_process` of tts_infer.py:57-64 (torchaudio defaults otherwise): waveform [..., n] -> [..., n_mels, frames]."""

    def __init__(self, sample_rate=24000, n_fft=1024, hop_length=256, n_mels=100, center=True, power=1.0, f_min=0.0, f_max=None):
        super().__init__()
        self.n_fft, self.hop_length, self.center, self.power = n_fft, hop_length, center, power
        self.register_buffer("window", torch.hann_window(n_fft, periodic=True), persistent=False)
        self.register_buffer("fb", melscale_fbanks(n_fft // 2 + 1, f_min, f_max if f_max is not None else sample_rate / 2.0,
                                                  n_mels, sample_rate), persistent=False)

    def forward(self, waveform):
        shape = waveform.shape
        x = waveform.reshape(-1, shape[-1])
        spec = torch.stft(x, self.n_fft, hop_length=self.hop_length, win_length=self.n_fft, window=self.window.to(x),
                          center=self.center, pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs()
        if self.power != 1.0:
            spec = spec.pow(self.power)
        mel = torch.matmul(spec.transpose(-1, -2), self.fb.to(spec)).transpose(-1, -2)
        return mel.reshape(shape[:-1] + mel.shape[-2:])


def reference_mel_prompt(waveform_24k, **kw):
    """tts_infer.py:57-67: log(clip(MelSpectrogram(...)(refer_audio24k), min=1e-7)) -> the `refer` tensor [B, 100, L]."""
    m = MelSpectrogram(**kw).to(waveform_24k.device)
    return torch.log(torch.clip(m(waveform_24k), min=1e-7))
