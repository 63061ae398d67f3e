#!/usr/bin/env python3
"""Stress of the in-launch hand-overs (GPU box): many sampler runs at shapes whose launches exceed the CU count (hand-overs planned by
rounds: kernels_gemm.hip gemm_handover_rounds) and at the bench shape (k_qkv_split's XCD-local exchange), checking after every run that
no wait timed out, no partner sat on a foreign XCD and the engine was not downgraded.

    python tools/handover_stress.py [runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from diff_vits_amd import synth
from diff_vits_amd.sampler import dpm_solver

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
ns = dpm_solver.NoiseScheduleVP("discrete", betas=torch.from_numpy(synth.make_betas()))
total = 0
for (B, T, L) in [(8, 1024, 256), (16, 1024, 256), (4, 2048, 256), (32, 512, 100), (8, 300, 150), (16, 99, 60)]:
    m, _ = bench.build_model(dev, "bf16x3")
    eng = m.hip_engine()
    x, cond, enc, mask = (torch.from_numpy(a).to(dev) for a in synth.make_inputs(B, 80, T, L, seed=5))
    native = dpm_solver.NativeUNetModel(m, cond, enc, mask)
    solver = dpm_solver.DPM_Solver(dpm_solver.model_wrapper(native, ns, model_type="x_start"), ns, algorithm_type="dpmsolver++")
    ref = None
    with torch.no_grad():
        for r in range(runs):
            out = solver.sample(x, steps=10, order=2, skip_type="time_uniform", method="multistep")
            ok = eng.wait()
            n_ho, bad = eng.handover_status()
            assert ok and not bad and not eng.handover_downgraded, (B, T, L, r, ok, bad, eng.handover_downgraded)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), "run %d differs from run 0" % r
            total += 10
    print("B=%-2d T=%-4d L=%-3d: %d runs x 10 evaluations, %d in-launch hand-over launches per forward, 0 time-outs, bit-identical" % (B, T, L, runs, n_ho), flush=True)
    del m, eng, native, solver
print("handover stress OK: %d evaluations" % total)
