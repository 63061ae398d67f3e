#!/usr/bin/env python3
"""GPU box: persistent per-XCD schedule (DVITS_PERSIST=1) against the per-launch schedule on the same inputs."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import diff_vits_amd  # noqa
from diff_vits_amd import synth
from diff_vits_amd.unet1d.unet_1d_condition import UNet1DConditionModel

KW = dict(in_channels=208, out_channels=80, block_out_channels=(128, 256, 384, 512), norm_num_groups=8, cross_attention_dim=128,
          attention_head_dim=8, addition_embed_type="text", resnet_time_scale_shift="scale_shift")


def build():
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in UNet1DConditionModel(**KW).state_dict().items()}
    sd = synth.make_state_dict(shapes)
    m = UNet1DConditionModel(**KW).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.cuda()


def run(B, T, L, persist, reps=20):
    os.environ["DVITS_PERSIST"] = "1" if persist else "0"
    m = build()
    x = torch.from_numpy(synth.normal(1, "x", (B, 80, T))).cuda()
    cond = torch.from_numpy(synth.normal(1, "c", (B, 128, T))).cuda()
    enc = torch.from_numpy(synth.normal(1, "e", (B, L, 128))).cuda()
    t = torch.full((B,), 500.0, device="cuda")
    eng = m.hip_engine()
    eng.sync_weights()
    eng.prepare(B, T, L)
    eng.set_cond(enc, None)
    y = eng.eval(x, cond, t).clone()
    n_ops, err = eng.persist_status()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.eval(x, cond, t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    return y.cpu().numpy(), n_ops, err, ms


if __name__ == "__main__":
    for B, T, L in ((8, 1024, 256), (8, 512, 64), (3, 512, 40)):
        a, n0, e0, ms0 = run(B, T, L, False)
        b, n1, e1, ms1 = run(B, T, L, True)
        rel = float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(a.astype(np.float64)))
        print("B=%d T=%d L=%d: per-launch %.3f ms | persistent %.3f ms (%d ops in one launch, error flag %d) | rel-L2 %.2e finite %s"
              % (B, T, L, ms0, ms1, n1, e1, rel, bool(np.isfinite(b).all())))
