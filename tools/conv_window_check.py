#!/usr/bin/env python3
"""Round 6 development check (GPU box): the convolution kernels outside round 5's window - padded row spaces (any T), grids
above the CU count - against the LDS-ring schedule of the same engine, with timings.

    python tools/conv_window_check.py [B,T,L ...]

For every shape and every environment variant: one forward (bit-repeatable, compared with the DVITS_CONV3=0 output), the number of
GEMM-kind operations on the convolution kernels and of in-launch GroupNorms, the hand-over flag, and ms per forward inside a
10-step DPM-Solver++ graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from diff_vits_amd import synth

VARIANTS = [("ring", {"DVITS_CONV3": "0"}),
            ("default", {}),
            ("min1", {"DVITS_CONV3_MIN_TILES": "1"}),
            ("rounds0", {"DVITS_GNX_ROUNDS": "0"}),
            ("min1_rounds0", {"DVITS_CONV3_MIN_TILES": "1", "DVITS_GNX_ROUNDS": "0"})]
if os.environ.get("CWC_VARIANTS"):
    keep = os.environ["CWC_VARIANTS"].split(",")
    VARIANTS = [v for v in VARIANTS if v[0] in keep]
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(3, 300, 150), (1, 300, 150), (16, 99, 60), (16, 1024, 256), (8, 1024, 256)]
dev = torch.device("cuda", 0)
for (B, T, L) in shapes:
    x, cond, enc, mask = (torch.from_numpy(a).to(dev) for a in synth.make_inputs(B, 80, T, L, seed=77))
    t = torch.linspace(900.0, 20.0, B, device=dev)
    ref = None
    for name, env in VARIANTS:
        os.environ.update(env)
        try:
            m, _ = bench.build_model(dev, "bf16x3")
            eng = m.hip_engine()
            eng.sync_weights()
            eng.prepare(B, T, L)
            eng.set_cond(enc, None)
            y = eng.eval(x, cond, t).clone()
            same = torch.equal(eng.eval(x, cond, t), y)
            torch.cuda.synchronize()
            n_ho, bad = eng.handover_status()
            rows = eng.profile_forward(x, cond, t)
            n_res = sum(1 for r in rows if r[0] == "gemm" and " resident" in r[3])
            n_gemm = sum(1 for r in rows if r[0] == "gemm")
            n_launch = eng.stats()[0]
            yc = y.double().cpu()
            if ref is None:
                ref = yc
            err = float((yc - ref).norm() / ref.norm())
            ms = bench.time_sampler(m, dev, B, T, L, "dpm", 10, runs=3) / 10.0
            torch.cuda.synchronize()
            n_ho2, bad2 = eng.handover_status()
            print("B=%-2d T=%-4d L=%-3d %-13s rel=%.2e repeat=%s conv-kernel ops %2d/%2d launches %3d gnx %2d timed_out %d/%d  %.3f ms/forward"
                  % (B, T, L, name, err, same, n_res, n_gemm, n_launch, n_ho, bad, bad2, ms), flush=True)
            del m, eng
        finally:
            for k in env:
                os.environ.pop(k, None)
