#!/usr/bin/env python3
"""Latency of ONE utterance (B = 1) through the native 30-step UniPC loop, for a few lengths (GPU box).
Usage: python tools/b1_latency.py [T ...]   (environment knobs such as DVITS_PERSIST=1 apply)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from diff_vits_amd import synth
from diff_vits_amd.sampler import uni_pc

dev = torch.device("cuda", 0)
m, _ = bench.build_model(dev, "bf16x3")
ns = uni_pc.NoiseScheduleVP("discrete", betas=torch.from_numpy(synth.make_betas()))
for T in [int(a) for a in sys.argv[1:]] or [192, 300, 512, 1024]:
    L = 150
    x, cond, enc, mask = (torch.from_numpy(a).to(dev) for a in synth.make_inputs(1, 80, T, L, seed=77))
    native = uni_pc.NativeUNetModel(m, cond, enc, mask)
    solver = uni_pc.UniPC(uni_pc.model_wrapper(native, ns, model_type="x_start"), ns, variant="bh2")
    with torch.no_grad():
        solver.sample(x, steps=30, order=2); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); solver.sample(x, steps=30, order=2); torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
    n_launch = m.hip_engine().stats()[0]
    print("B=1 T=%d L=%d: %.1f ms per 30-step run (%.2f ms per forward, %d launches per forward), persist ops %s"
          % (T, L, sorted(ts)[2], sorted(ts)[2] / 30, n_launch, m.hip_engine().persist_status()))
