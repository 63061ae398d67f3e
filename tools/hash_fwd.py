#!/usr/bin/env python3
"""Hash of the denoiser output at three shapes (GPU box): two builds whose change is meant to be bit-neutral must print the same
lines.  Usage: python tools/hash_fwd.py   |   DVITS_LIB_FILE=diff-vits_amd/libdvits_hip_<other>.so python tools/hash_fwd.py"""
import sys, hashlib, torch
sys.path.insert(0, '.')
import bench
from diff_vits_amd import synth
dev = torch.device("cuda", 0)
m, _ = bench.build_model(dev, "bf16x3")
for (B, T, L) in [(8, 1024, 256), (2, 300, 77), (1, 256, 128)]:
    x, cond, enc, mask = (torch.from_numpy(a).to(dev) for a in synth.make_inputs(B, 80, T, L, seed=5))
    t = torch.linspace(900., 30., B, device=dev)
    with torch.no_grad():
        y = m(torch.cat([x, cond], 1), t, enc, encoder_attention_mask=mask).sample
    torch.cuda.synchronize()
    print(B, T, L, hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16], float(y.abs().mean()))
