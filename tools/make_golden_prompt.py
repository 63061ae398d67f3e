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


if __name__ == "__main__":
    main()
