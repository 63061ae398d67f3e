#!/usr/bin/env python3
"""Native sampler loop with the batched time-embedding chain against the per-step chain (child process with
DVITS_TEMB_BATCH=0), for a few (B, steps): the two must agree bit for bit.  GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def run(B, steps, T=64, L=24):
    import bench
    from diff_vits_amd import synth
    from diff_vits_amd.sampler import dpm_solver
    dev = torch.device("cuda", 0)
    m, _ = bench.build_model(dev, "bf16x3")
    x, cond, enc, mask = (torch.from_numpy(a).to(dev) for a in synth.make_inputs(B, 80, T, L, seed=3))
    ns = dpm_solver.NoiseScheduleVP("discrete", betas=torch.from_numpy(synth.make_betas()))
    native = dpm_solver.NativeUNetModel(m, cond, enc, mask)
    fn = dpm_solver.model_wrapper(native, ns, model_type="x_start")
    with torch.no_grad():
        y = dpm_solver.DPM_Solver(fn, ns, algorithm_type="dpmsolver++").sample(x, steps=steps, order=2, method="multistep")
    return y.cpu().numpy()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        B, steps = int(sys.argv[2]), int(sys.argv[3])
        np.save(sys.argv[4], run(B, steps))
        sys.exit(0)
    for B, steps in ((2, 8), (2, 10), (1, 16), (8, 2), (3, 5), (16, 2)):
        a = run(B, steps)
        f = "/tmp/temb_ref_%d_%d.npy" % (B, steps)
        subprocess.run([sys.executable, __file__, "child", str(B), str(steps), f], env=dict(os.environ, DVITS_TEMB_BATCH="0"), check=True)
        b = np.load(f)
        print("B=%d steps=%d (rows %d): max |diff| %.3e  equal=%s" % (B, steps, B * steps, float(np.abs(a - b).max()), bool((a == b).all())))
