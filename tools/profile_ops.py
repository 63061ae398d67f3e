#!/usr/bin/env python3
"""Per-launch table of one denoiser forward at the bench shape (GPU box): kernel family, shape,
algorithmic FLOPs, HIP-event time, achieved TFLOP/s.  Aggregates identical shapes."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from diff_vits_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--prompt", type=int, default=256)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--summary", action="store_true", help="per-kernel-family totals only")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev, a.precision)
    x, cond, enc, mask = (torch.from_numpy(v).to(dev) for v in synth.make_inputs(a.batch, 80, a.frames, a.prompt))
    eng = model.hip_engine()
    eng.prepare(a.batch, a.frames, a.prompt)
    eng.set_cond(enc, None)
    t = torch.full((a.batch,), 500.0, device=dev)
    eng.profile_forward(x, cond, t)
    agg = {}
    for _ in range(a.reps):
        for kind, fl, ms, desc in eng.profile_forward(x, cond, t):
            e = agg.setdefault((kind, desc), [0, fl, 0.0])
            e[0] += 1
            e[2] += ms
    rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
    tot = sum(v[2] for v in agg.values()) / a.reps
    print("total %.3f ms per forward (event-timed, eager)  DVITS_GEMM_CFG=%s" % (tot, os.environ.get("DVITS_GEMM_CFG", "default")))
    fam = {}
    for (kind, desc), (n, fl, ms) in agg.items():
        f = fam.setdefault(kind, [0, 0.0, 0.0])
        f[0] += n // a.reps
        f[1] += ms / a.reps
        f[2] += fl * (n // a.reps)
    for k, (n, ms, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("  %-12s launches=%4d  %8.3f ms  %7.1f TF/s" % (k, n, ms, fl / (ms * 1e-3) / 1e12 if fl else 0.0))
    if a.summary:
        return
    for (kind, desc), (n, fl, ms) in rows:
        per = ms / n
        print("%-11s %-62s x%-3d %8.1f us  %7.1f TF/s  %5.1f%%" % (kind, desc, n // a.reps, per * 1e3,
              fl / (per * 1e-3) / 1e12 if fl else 0.0, 100 * (ms / a.reps) / tot))


if __name__ == "__main__":
    main()
