#!/usr/bin/env python3
"""Headline benchmark: mel-frames/sec of the 50-step DPM-Solver++(2M) sampler on synthetic
(B=8 per GPU, C=80, T=1024, L=256) — BASELINE.json configs[1] — on N MI355X GPUs of one node.

A "step" is one complete sampler run (50 denoiser evaluations + 50 fused updates) over one
batch per GPU.  One JSON line is printed by rank 0 (contract in the task description), with
`roofline` (dominant kernel = the implicit-GEMM family, timed live with HIP events) and
`cpu_baseline` (the oracle restatement timed on the host cores; bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (one per GPU, RCCL
    rendezvous on 127.0.0.1) and exit with the worst of their codes.  Runs before torch / the HIP library are imported,
    so the parent never touches a GPU; the children are fresh processes, not re-execs of an initialised one.
    The children are polled: as soon as one exits non-zero the siblings are terminated (they would otherwise sit in
    `init_process_group` / a collective until the RCCL time-out), so a rank failure returns within seconds."""
    import signal
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DVITS_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))

    def stop_all(sig=signal.SIGTERM):
        for q in procs:
            if q.poll() is None:
                try:
                    q.send_signal(sig)
                except OSError:
                    pass

    rc = 0
    deadline = time.time() + float(os.environ.get("DVITS_BENCH_JOB_TIMEOUT_S", "3600"))
    try:
        live = list(procs)
        while live:
            for q in list(live):
                code = q.poll()
                if code is None:
                    continue
                live.remove(q)
                if code != 0:
                    rc = max(rc, abs(code) or 1)
            if rc != 0 and live:                       # a rank died: do not wait for the others' collective time-outs
                sys.stderr.write("bench: a rank exited with code %d - stopping the other %d rank(s)\n" % (rc, len(live)))
                stop_all()
                t_kill = time.time() + 10.0
                while any(q.poll() is None for q in live) and time.time() < t_kill:
                    time.sleep(0.1)
                stop_all(signal.SIGKILL)
                for q in live:
                    q.wait()
                break
            if time.time() > deadline:
                sys.stderr.write("bench: job time-out - stopping all ranks\n")
                stop_all(signal.SIGKILL)
                rc = max(rc, 124)
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        stop_all(signal.SIGKILL)
        rc = 130
    sys.exit(rc)


def _early_gpus():
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            return int(sys.argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _early_gpus() > 1:
    spawn_ranks(_early_gpus())

import numpy as np  # noqa: E402
import torch  # noqa: E402

import diff_vits_amd  # noqa: E402,F401
from diff_vits_amd import shard, synth  # noqa: E402
from diff_vits_amd.engine import HandoverLost  # noqa: E402
from diff_vits_amd.sampler import dpm_solver  # noqa: E402
from diff_vits_amd.unet1d.unet_1d_condition import UNet1DConditionModel  # noqa: E402

UNET_KW = dict(in_channels=208, out_channels=80, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
               cross_attention_dim=128, attention_head_dim=8, addition_embed_type="text",
               resnet_time_scale_shift="scale_shift")
PEAK_TFLOPS = {"bf16x3": 2500.0 / 3.0, "bf16": 2500.0}     # dense bf16 MFMA peak (MI355X_MICROARCH.md); x3 = 3 products
PARITY_T = 500.0          # timestep of the forward whose output is compared with the oracle's in the bench line
PARITY_TOL = 1e-3         # BASELINE.json north_star: UNet output <= 1e-3 relative error
# --dry-run-cpu: the rank path of this script (spawn -> env -> init_process_group -> shard.sharded_sample -> MAX-reduce ->
# rank-0 JSON) without a GPU: gloo, the package's explicit torch backend, a tiny denoiser configuration
DRY_KW = dict(in_channels=24, out_channels=8, block_out_channels=(32, 64, 96, 128), norm_num_groups=8,
              cross_attention_dim=32, attention_head_dim=8, addition_embed_type="text",
              resnet_time_scale_shift="scale_shift", addition_embed_type_num_heads=8)


def flops_model(B, T, L):
    """Algorithmic FLOPs of one denoiser forward, closed form fitted in SURVEY.md §8(d)."""
    return B * (31.67e6 * T + 4349.0 * T * T + 7286.0 * T * L + 2.23e6 * L + 0.04e9)


def build_model(device, precision, dry=False):
    kw = DRY_KW if dry else UNET_KW
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in UNet1DConditionModel(**kw).state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(shapes, seed=1234).items()}
    m = UNet1DConditionModel(backend="torch" if dry else "hip", **kw).eval()
    m.load_state_dict(sd)
    m = m.to(device)
    if not dry:
        m.hip_engine(precision)
    return m, sd


def enc_dim_of(dry):
    return DRY_KW["cross_attention_dim"] if dry else UNET_KW["cross_attention_dim"]


FAMILY_KERNELS = {
    "gemm": "k_gemm<*> + k_conv3 / k_conv3s / k_conv3u (implicit-GEMM conv1d / linear: every launch of the engine's GEMM kind)",
    "chain": "k_chain2<*> + k_chain_ff + k_ff_split<*> (row-block chains of the transformer blocks: every launch of the engine's chain kind)",
    "attn": "k_attention_frag<*> + k_attention<*> (flash attention: every launch of the engine's attention kind)",
}
# engine kind -> the families of tools/pmc_roofline.py's JSON that make it up
PMC_FAMILIES = {"gemm": ("gemm",), "chain": ("chain", "ff_split"), "attn": ("attention",)}


def roofline_families(live, pmc, peak, engine_counted_gflop):
    """The MFMA kernel families of one forward side by side, and which of them the headline `roofline` fields describe.

    live: {kind: (launches_per_forward, avg_launch_us, gflop_per_forward)} - launch durations measured live (the family's
          launches of the schedule replayed back to back between one HIP event pair on the launch stream)
    pmc:  the committed rocprofv3 JSON of this build (tools/pmc_roofline.py) or None
    Returns (families, dominant): every family carries its achieved TFLOP/s and fraction of `peak` live AND by rocprofv3's
    per-kernel average of the committed kernel trace, its share of the forward's kernel time and of its FLOPs; `dominant` is the
    family with the most kernel time - the one `roofline.kernel / achieved / frac / frac_rocprofv3` describe.
    The families' FLOPs sum to the engine's own count (dv_unet_stats): asserted."""
    fam = {}
    t_all = sum(n * us for n, us, _ in live.values())
    for kind, (n, us, gf) in live.items():
        ach = gf * 1e9 / (us * 1e-6 * n) / 1e12 if n and us > 0 else 0.0
        f = {"kernel": FAMILY_KERNELS.get(kind, kind), "launches_per_forward": n, "avg_launch_us": us, "gflop_per_forward": gf,
             "ms_per_forward_back_to_back": us * n * 1e-3, "time_share": (n * us / t_all) if t_all > 0 else 0.0,
             "flop_share": gf / engine_counted_gflop if engine_counted_gflop > 0 else 0.0,
             "achieved": ach, "frac": ach / peak, "rocprofv3": None, "frac_rocprofv3": None}
        parts = [pmc[k] for k in PMC_FAMILIES.get(kind, ()) if pmc and k in pmc and pmc[k].get("launches")]
        if parts:
            ln = sum(q["launches"] for q in parts)
            rp_us = sum(q["launches"] * q["avg_us_kernel_trace"] for q in parts) / ln
            # (the trace holds several forwards: its launch count is a multiple of the family's launches per forward)
            ach_rp = gf * 1e9 / (rp_us * 1e-6 * n) / 1e12 if n else 0.0
            f["rocprofv3"] = {"avg_launch_us": rp_us, "launches_in_trace": ln, "achieved": ach_rp,
                              "hbm_gbps": sum(q["launches"] * q["avg_us_kernel_trace"] * q["hbm_gbps"] for q in parts) / (ln * rp_us),
                              "hbm_bytes_per_launch": sum(q["launches"] * q["hbm_bytes_per_launch"] for q in parts) / ln,
                              "mfma_util": sum(q["launches"] * q["avg_us_kernel_trace"] * q["mfma_util"] for q in parts) / (ln * rp_us)}
            f["frac_rocprofv3"] = ach_rp / peak
        fam[kind] = f
    total = sum(f["gflop_per_forward"] for f in fam.values())
    assert abs(total - engine_counted_gflop) <= 1e-6 * max(engine_counted_gflop, 1.0), \
        "family FLOPs %.3f GF do not sum to the engine's count %.3f GF" % (total, engine_counted_gflop)
    dominant = max(fam, key=lambda k: fam[k]["ms_per_forward_back_to_back"])
    return fam, dominant


def pmc_identity_ok(pj):
    """The committed PMC figures belong to ONE build of the kernels: tools/pmc_roofline.py stamps them with the library's
    dv_version() (which carries a hash of csrc/).  They are reported only while that still is the loaded library."""
    from diff_vits_amd import _lib
    return isinstance(pj, dict) and pj.get("build", {}).get("dv_version") == _lib.lib().dv_version().decode()


def cpu_baseline(sd, B, T, L, solver_steps, sample_steps=2):
    """Oracle (CPU restatement of the reference, oracle/) on the host cores (SURVEY.md §8d "CPU reference timing").
    The thread count is swept over {8, 16, 32, 64, all} capped by this process's CPU affinity (128 threads on a
    128-core host measured 3.6x SLOWER than 8 in round 1: oversubscription); per count one warm-up forward and the
    median of 3 forwards at the bench shape.  The best count then runs `sample_steps` solver steps (median of 3) and
    that time is scaled to `solver_steps` (NFE == steps for the multistep solver)."""
    import statistics
    from oracle import sampler_ref, unet_ref
    cfg = unet_ref.default_config()
    x, cond, enc, mask = map(torch.from_numpy, synth.make_inputs(B, 80, T, L, seed=1234))
    betas = torch.from_numpy(synth.make_betas())
    model = unet_ref.diffusion_model_fn(sd, cfg, cond, enc, mask)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # ascending; "all" only on hosts with <= 64 CPUs (one forward on 256 threads of a 256-CPU box took 134 s in round 2:
    # oversubscribed intra-op pools), and the sweep stops as soon as a count is clearly past the optimum
    counts = sorted({min(c, avail) for c in (8, 16, 24, 32, 48, 64)} | ({avail} if avail <= 64 else set()))
    t_in = torch.full((B,), PARITY_T)
    t_all0 = time.perf_counter()
    sweep = {}
    y_ref = None
    with torch.no_grad():
        for c in counts:
            torch.set_num_threads(c)
            t0 = time.perf_counter()
            y_ref = model(x, t_in)                                  # warm-up (thread pool, allocator, oneDNN primitives)
            warm = time.perf_counter() - t0
            if sweep and warm > 4.0 * min(sweep.values()):          # far past the optimum: do not spend the budget here
                sweep[c] = warm
                break
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                model(x, t_in)
                ts.append(time.perf_counter() - t0)
            sweep[c] = statistics.median(ts)
            # past the optimum (two counts in a row slower than the best so far) / bounded
            worse = [k for k in sweep if sweep[k] > 1.3 * min(sweep.values())]
            if len(worse) >= 2 or time.perf_counter() - t_all0 > 40.0:
                break
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            sampler_ref.dpm_solver_pp_sample(model, betas, x, sample_steps, 2)
            runs.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all0 > 90.0:
                break
    dt = statistics.median(runs)
    full = dt * solver_steps / sample_steps
    return {"value": B * T / full, "unit": "mel-frames/s", "cores": best, "kind": "port",
            "sample": "median of %d runs of %d of the %d DPM-Solver++ steps at B=%d,T=%d,L=%d on torch-CPU fp32 with %d "
                      "threads (best of the sweep; %d CPUs in the affinity mask), scaled x%.1f"
                      % (len(runs), sample_steps, solver_steps, B, T, L, best, avail, solver_steps / sample_steps),
            "seconds_sampled": time.perf_counter() - t_all0,
            "thread_sweep_forward_s": {str(k): round(v, 4) for k, v in sweep.items()}}, y_ref


def lifecycle_extras(dev, precision):
    """Engine life-cycle costs around the timed region: weight packing + planning for a first shape,
    re-planning for another shape (packed weights are kept), hipGraph capture of a 30-step UniPC loop, and the latency
    of ONE utterance at a length that is not a multiple of anything (B=1, T=300, L=150: the reference's default call)."""
    from diff_vits_amd.sampler import uni_pc
    out = {}
    m, _ = build_model(dev, precision)
    eng = m.hip_engine()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng._prepared = None; eng.prepare(1, 256, 150); torch.cuda.synchronize()
    out["replan_ms_weights_kept"] = 1e3 * (time.perf_counter() - t0)
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in UNet1DConditionModel(**UNET_KW).state_dict().items()}
    m2 = UNet1DConditionModel(backend="hip", **UNET_KW).eval()
    m2.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(shapes, seed=1234).items()})
    m2 = m2.to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e2 = m2.hip_engine(precision); e2.prepare(1, 300, 150); torch.cuda.synchronize()
    out["first_prepare_ms_incl_weight_upload_and_packing"] = 1e3 * (time.perf_counter() - t0)
    x, cond, enc, mask = (torch.from_numpy(a).to(dev) for a in synth.make_inputs(1, 80, 300, 150, seed=77))
    ns = uni_pc.NoiseScheduleVP("discrete", betas=torch.from_numpy(synth.make_betas()))
    native = uni_pc.NativeUNetModel(m2, cond, enc, mask)
    solver = uni_pc.UniPC(uni_pc.model_wrapper(native, ns, model_type="x_start"), ns, variant="bh2")
    with torch.no_grad():
        t0 = time.perf_counter(); solver.sample(x, steps=30, order=2); torch.cuda.synchronize()
        out["first_run_ms_incl_graph_capture"] = 1e3 * (time.perf_counter() - t0)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); solver.sample(x, steps=30, order=2); torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
    out["b1_T300_L150_unipc30_latency_ms"] = sorted(ts)[len(ts) // 2]
    out["b1_T300_mel_frames_per_s"] = 300.0 / (out["b1_T300_L150_unipc30_latency_ms"] * 1e-3)
    # utterances of recurring lengths: schedules (engine LRU) and captured graphs (per-shape sampler plans) are reused
    lat = {}
    with torch.no_grad():
        for rnd in range(2):
            for T2, L2 in ((300, 150), (192, 60), (416, 200)):
                xa, ca, ea, ma = (torch.from_numpy(a).to(dev) for a in synth.make_inputs(1, 80, T2, L2, seed=78))
                native.cond, native.enc, native.mask = ca, ea, ma
                torch.cuda.synchronize()
                t0 = time.perf_counter(); solver.sample(xa, steps=30, order=2); torch.cuda.synchronize()
                lat.setdefault("T%d" % T2, []).append(round(1e3 * (time.perf_counter() - t0), 2))
    out["b1_unipc30_latency_ms_first_then_repeat"] = lat
    # GEMMs that finish their consumer's GroupNorm in the epilogue (in-launch hand-over); the flag must stay 0
    n_ho, bad = eng.handover_status()
    n_ho2, bad2 = e2.handover_status()
    out["in_epilogue_groupnorm_gemms"] = {"B1_T256": n_ho, "B1_T300": n_ho2, "timed_out": int(bool(bad or bad2))}
    return out


def time_sampler(model, dev, B, T, L, kind, steps, runs=2, seed=4321):
    """ms per complete sampler run (median of `runs` after one warm-up run that plans the shape and captures the graph)."""
    from diff_vits_amd.sampler import uni_pc
    x, cond, enc, mask = (torch.from_numpy(a).to(dev) for a in synth.make_inputs(B, 80, T, L, seed=seed))
    betas = torch.from_numpy(synth.make_betas())
    if kind == "unipc":
        ns = uni_pc.NoiseScheduleVP("discrete", betas=betas)
        native = uni_pc.NativeUNetModel(model, cond, enc, mask)
        solver = uni_pc.UniPC(uni_pc.model_wrapper(native, ns, model_type="x_start"), ns, variant="bh2")
        run = lambda: solver.sample(x, steps=steps, order=2)                                           # noqa: E731
    else:
        ns = dpm_solver.NoiseScheduleVP("discrete", betas=betas)
        native = dpm_solver.NativeUNetModel(model, cond, enc, mask)
        solver = dpm_solver.DPM_Solver(dpm_solver.model_wrapper(native, ns, model_type="x_start"), ns, algorithm_type="dpmsolver++")
        run = lambda: solver.sample(x, steps=steps, order=2, skip_type="time_uniform", method="multistep")   # noqa: E731
    ts = []
    with torch.no_grad():
        out = run()
        torch.cuda.synchronize()
        for _ in range(runs):
            t0 = time.perf_counter(); out = run(); torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
    assert torch.isfinite(out).all()
    return sorted(ts)[len(ts) // 2]


def other_configs(dev, model, sd, precision, S, y_ref):
    """The single-GPU BASELINE.json configurations beside the headline, a few seconds in all:
    config 2 in the fast precision mode (one bf16 product; outside the 1e-3 parity budget - its error is reported beside
    it), config 4 (UniPC-bh2, 20 steps, B=1, T=2048 long-form) and the headline shapes at 16 utterances per GPU."""
    out = {}
    peak = PEAK_TFLOPS[precision]
    # config 4: UniPC (sampler/uni_pc.py), 20 steps, order 2, one utterance of 2048 frames
    ms = time_sampler(model, dev, 1, 2048, 256, "unipc", 20)
    out["config4_unipc20_T2048_B1"] = {"value": 2048.0 / (ms * 1e-3), "unit": "mel-frames/s", "ms_per_run": ms,
                                       "frac_of_peak": 20 * flops_model(1, 2048, 256) / (ms * 1e-3) / 1e12 / peak,
                                       "precision": precision}
    ms = time_sampler(model, dev, 16, 1024, 256, "dpm", S)
    out["b16"] = {"value": 16 * 1024.0 / (ms * 1e-3), "unit": "mel-frames/s", "ms_per_run": ms,
                  "frac_of_peak": S * flops_model(16, 1024, 256) / (ms * 1e-3) / 1e12 / peak, "precision": precision}
    # the batch-independent floor, driver-timed: ms per forward inside a 10-step graph at B = 1 / 2 / 4 / 16
    sweep = {}
    for Bs in (1, 2, 4, 16):
        ms = time_sampler(model, dev, Bs, 1024, 256, "dpm", 10, runs=3)
        sweep["B%d" % Bs] = {"ms_per_forward": ms / 10.0, "mel_frames_per_s_at_%d_steps" % S: Bs * 1024.0 / (ms / 10.0 * S * 1e-3)}
    out["batch_sweep"] = sweep
    if precision == "bf16x3":
        fast, _ = build_model(dev, "bf16")
        ms = time_sampler(fast, dev, 8, 1024, 256, "dpm", S)
        fm = {"value": 8 * 1024.0 / (ms * 1e-3), "unit": "mel-frames/s", "ms_per_run": ms,
              "frac_of_peak": S * flops_model(8, 1024, 256) / (ms * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"],
              "precision": "bf16 (one product per contraction; NOT the parity mode)"}
        if y_ref is not None:
            xo, co, eo, mo = (torch.from_numpy(a).to(dev) for a in synth.make_inputs(8, 80, 1024, 256, seed=1234))
            with torch.no_grad():
                y = fast(torch.cat([xo, co], 1), torch.full((8,), PARITY_T, device=dev), eo, encoder_attention_mask=mo).sample
            y = y.double().cpu()
            yr = y_ref.double()
            fm["unet_rel_l2"] = float((y - yr).norm() / yr.norm())
            fm["unet_max_abs_rel"] = float((y - yr).abs().max() / yr.abs().max())
        out["bf16_fast_mode"] = fm
        del fast
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed sampler runs")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--prompt", type=int, default=256)
    ap.add_argument("--solver-steps", type=int, default=50)
    ap.add_argument("--precision", default=os.environ.get("DVITS_PRECISION", "bf16x3"), choices=["bf16x3", "bf16"])
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DVITS_BENCH_STREAMS", "1")),
                    help="split the per-GPU batch into this many concurrent sub-batches (own engine + HIP stream each)")
    ap.add_argument("--keep-handover", action="store_true", default=os.environ.get("DVITS_BENCH_KEEP_HANDOVER") == "1",
                    help="with --streams k: the engines keep their in-launch hand-overs (plan each with DVITS_CU_BUDGET = CUs / k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip extra.{bf16_fast_mode, config4_unipc20_T2048_B1, b16} (the other single-GPU BASELINE configurations)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="GPU-less rehearsal of the multi-rank path: gloo, torch backend, tiny configuration (tests)")
    ap.add_argument("--dist-timeout", type=float, default=float(os.environ.get("DVITS_DIST_TIMEOUT_S", "180")),
                    help="seconds a rank waits in the rendezvous / a collective before it gives up")
    args = ap.parse_args()
    dry = args.dry_run_cpu
    if dry:      # small enough for 8 ranks on a few CPU cores; explicit flags still win
        defaults = {"batch": 2, "frames": 24, "prompt": 10, "solver_steps": 4}
        for k, v in defaults.items():
            if getattr(args, k) == ap.get_default(k):
                setattr(args, k, v)
        args.no_roofline = args.no_cpu_baseline = args.no_other_configs = True
        torch.set_num_threads(1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d (run `python bench.py --gpus N`, which starts the N "
                         "ranks itself, or torchrun --nproc-per-node N bench.py --gpus N)" % (args.gpus, world))
    if not dry and torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d: only %d GPU(s) visible" % (world, torch.cuda.device_count()))
    if os.environ.get("DVITS_BENCH_FAIL_RANK") == str(rank):     # fault injection for the sibling-kill test
        raise SystemExit("bench: injected failure on rank %d" % rank)
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        import datetime
        import torch.distributed as dist
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)

    B, T, L, S = args.batch, args.frames, args.prompt, args.solver_steps
    NS = max(1, args.streams)
    if B % NS != 0:
        raise SystemExit("--batch must be divisible by --streams")
    if dry and NS != 1:
        raise SystemExit("--dry-run-cpu runs one sub-batch per rank")
    model, sd = build_model(dev, args.precision, dry)
    replicas = [model] + [build_model(dev, args.precision)[0] for _ in range(NS - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)] if NS > 1 else [None]
    if NS > 1 and not args.keep_handover:   # engines driven side by side on one device: no in-launch GroupNorm hand-over (dv_unet_set_exclusive)
        for r in replicas:
            r.hip_engine(args.precision).set_exclusive(False)

    # synthetic inputs: this rank's noise/content shard is generated locally (zero traffic); the
    # conditioning of the whole job lives on rank 0 and is broadcast over RCCL before every run
    CM = DRY_KW["out_channels"] if dry else 80       # mel channels
    in_kw = dict(cond_channels=DRY_KW["in_channels"] - CM, enc_dim=DRY_KW["cross_attention_dim"]) if dry else {}
    x_np, cond_np, _, _ = synth.make_inputs(B, CM, T, L, seed=1234 + rank, **in_kw)
    x_T = torch.from_numpy(x_np).to(dev)
    cond = torch.from_numpy(cond_np).to(dev)
    G = world * B
    if rank == 0:
        enc_all = np.concatenate([synth.make_inputs(B, CM, 8, L, seed=1234 + r, **in_kw)[2] for r in range(world)])
        enc_g = torch.from_numpy(enc_all).to(dev)
    else:
        enc_g = torch.empty((G, L, enc_dim_of(dry)), device=dev)
    mask_g = torch.ones((G, L), dtype=torch.bool, device=dev)

    ns = dpm_solver.NoiseScheduleVP("discrete", betas=torch.from_numpy(synth.make_betas()))
    state = {}

    def run_sub(i, x, c, enc, mask):
        key = ("native", i)
        native = state.get(key)
        if native is None:
            native = state[key] = dpm_solver.NativeUNetModel(replicas[i], c, enc, mask)
            fn = dpm_solver.model_wrapper(native, ns, model_type="x_start")
            state[("solver", i)] = dpm_solver.DPM_Solver(fn, ns, algorithm_type="dpmsolver++")
        native.cond, native.enc, native.mask = c, enc, mask
        try:
            return state[("solver", i)].sample(x, steps=S, order=2, skip_type="time_uniform", method="multistep")
        except HandoverLost:
            # lazy hand-over verification (diff_vits_amd/engine.py): an EARLIER run's timed-out in-launch hand-over is noticed by
            # this call - the engine has recovered (fallback schedule); repeat, the line below reports the downgrade and the bench
            # refuses the number
            return state[("solver", i)].sample(x, steps=S, order=2, skip_type="time_uniform", method="multistep")

    def run_local(x, c, enc, mask):
        if NS == 1:
            return run_sub(0, x, c, enc, mask)
        bs = B // NS
        cur = torch.cuda.current_stream()
        outs = []
        for i in range(NS):
            sl = slice(i * bs, (i + 1) * bs)
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs.append(run_sub(i, x[sl].contiguous(), c[sl].contiguous(), enc[sl].contiguous(), mask[sl].contiguous()))
        for i in range(NS):
            cur.wait_stream(streams[i])
        return torch.cat(outs, dim=0)

    def one_run():
        return shard.sharded_sample(run_local, x_T, cond, enc_g, mask_g)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        if not dry:
            torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = one_run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = one_run()
        barrier()
        dt = time.perf_counter() - t0
    rank_ms = [1e3 * dt / args.steps]                # per-rank ms per step: a straggler shows up as max >> min
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = torch.empty((world,), device=dev, dtype=torch.float64)
        torch.distributed.all_gather_into_tensor(every, tt)
        rank_ms = [1e3 * float(v) / args.steps for v in every.tolist()]
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(out).all()
    assert out.shape[0] == G, "the gathered mels must cover the global batch"
    # Which schedule did every rank time?  (GEMMs that finish a GroupNorm in their epilogue, time-out flag, downgraded to the
    # separate-GroupNorm schedule): a rank whose in-launch hand-over timed out - RCCL kernels on a side stream are exactly the
    # foreign work that can trip the bounded poll - has repeated runs on the slower fallback schedule; the line must say so.
    ho_ranks = None
    if not dry:
        eng0 = model.hip_engine()
        mine = [0, 0, 0]
        for r_ in replicas:                          # (--streams k: every sub-batch's engine)
            e_ = r_.hip_engine()
            lost = not e_.wait()                     # (the stream has drained above: verifies the last runs, recovers if one was lost)
            n_ho_r, bad_r = e_.handover_status()
            mine = [mine[0] + int(n_ho_r), int(mine[1] or bool(bad_r) or lost), int(mine[2] or bool(e_.handover_downgraded))]
        if world > 1:
            hv = torch.tensor(mine, device=dev, dtype=torch.int64)
            hall = torch.empty((world, 3), device=dev, dtype=torch.int64)
            torch.distributed.all_gather_into_tensor(hall, hv)
            ho_ranks = hall.tolist()
        else:
            ho_ranks = [mine]

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    frames_per_s = G * T * args.steps / dt
    result = {
        "metric": "mel_frames_per_sec_50step_dpmsolver", "value": frames_per_s, "unit": "mel-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "%s (bf16 MFMA operands%s, fp32 accumulate / fp32 activations)" % (
            args.precision, " split hi+lo, 3 products" if args.precision == "bf16x3" else ""),
        "data": "synthetic",
        "config": {"workload": "B=%d/GPU, C=%d, T=%d, L=%d, %d-step DPM-Solver++(2M) multistep, UNet1DConditionModel "
                               "%s, seeded random-init weights" % (B, CM, T, L, S, tuple((DRY_KW if dry else UNET_KW)["block_out_channels"])),
                   "global_batch": G, "concurrent_sub_batches_per_gpu": NS,
                   "rccl_world_size": torch.distributed.get_world_size() if world > 1 else 1, "parallelism": "dp%d (batch-sharded, RCCL broadcast of conditioning + all-gather of mels)" % world},
        "per_rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "ranks": [round(v, 3) for v in rank_ms]},
    }
    if ho_ranks is not None:
        result["per_rank_in_epilogue_groupnorm"] = {"gemms": [r[0] for r in ho_ranks], "timed_out": [r[1] for r in ho_ranks],
                                                    "downgraded_to_separate_groupnorm": [r[2] for r in ho_ranks]}
        if any(r[1] or r[2] for r in ho_ranks):
            print(json.dumps(result), flush=True)
            if world > 1:                            # (the other ranks have already left the group and returned)
                torch.distributed.destroy_process_group()
            raise SystemExit("bench: an in-kernel GroupNorm hand-over timed out on rank(s) %s (is the GPU shared?) - not the default "
                             "schedule's number" % [i for i, r in enumerate(ho_ranks) if r[1] or r[2]])
    if dry:
        result["dry_run_cpu"] = True                 # a rehearsal of the rank path on CPU (gloo, torch backend): not a measurement
        result["dtype"] = "fp32 (torch backend, CPU dry run)"
        result["config"]["parallelism"] = "dp%d (batch-sharded, gloo broadcast of conditioning + all-gather of mels)" % world

    # ---- roofline of the dominant kernel family (implicit GEMM), timed live with HIP events ----
    if not args.no_roofline:
        eng = model.hip_engine()
        bsub = B // NS
        t_dev = torch.full((bsub,), 500.0, device=dev)
        xs_, cs_ = x_T[:bsub].contiguous(), cond[:bsub].contiguous()
        reps = 5
        agg = {}
        with torch.no_grad():
            eng.profile_forward(xs_, cs_, t_dev)                # warm
            for _ in range(reps):
                for kind, fl, ms, _ in eng.profile_forward(xs_, cs_, t_dev):
                    a = agg.setdefault(kind, [0, 0.0, 0.0])
                    a[0] += 1
                    a[1] += fl
                    a[2] += ms
        n_launch, flops_fwd = eng.stats()
        peak = PEAK_TFLOPS[args.precision]
        # Average launch duration per MFMA kernel family: the family's launches of the schedule replayed back to back between ONE
        # event pair on the launch stream (no per-launch event overhead; this is the figure rocprofv3's per-kernel average must
        # match); the per-launch event pairs above (~+2.5 us each) give the per-kind split and are reported beside it.  FLOPs per
        # family: the engine's own per-operation counts (2 M N K per contraction, 4 B H Tq Tk d per attention; a chain launch
        # counts every contraction and the attention inside it), which sum to dv_unet_stats' total.
        live = {}
        for kind in ("gemm", "chain", "attn"):
            if kind in agg:
                us_k, n_k = eng.time_family(kind, reps=10)
                live[kind] = (n_k, us_k, agg[kind][1] / reps / 1e9)
        for kind, v in agg.items():                  # (kinds without an MFMA: no FLOPs are counted for them)
            if kind not in live and v[1] > 0:
                live[kind] = (v[0] // reps, 1e3 * v[2] / v[0], v[1] / reps / 1e9)
        fwd_ms = 1e3 * dt / args.steps / S
        # HBM-side bytes per launch, HBM GB/s and MFMA utilisation: PMC counters cannot be read from inside the process, so
        # these are the figures of the committed rocprofv3 --pmc passes over this same command (tools/pmc_roofline.py)
        traffic_src = None
        pmc_families = None
        # (the newest committed evidence file: profiles/rNN_pmc_roofline.json)
        import glob
        cands = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_pmc_roofline.json")))
        tpath = cands[-1] if cands else ""
        tname = "profiles/" + os.path.basename(tpath)
        pj = None
        if tpath and args.precision == "bf16x3" and (B, T, L) == (8, 1024, 256):
            with open(tpath) as f:
                pj = json.load(f)
        if pj is not None and not pmc_identity_ok(pj):
            from diff_vits_amd import _lib
            traffic_src = ("stale: %s was collected on build %r, the loaded library is %r - PMC "
                           "fields withheld" % (tname, pj.get("build", {}).get("dv_version"), _lib.lib().dv_version().decode()))
            pj = None
        elif pj is not None:
            pmc_families = {k: {kk: v[kk] for kk in ("launches", "avg_us_kernel_trace", "hbm_bytes_per_launch", "hbm_gbps", "mfma_util")}
                            for k, v in pj.items() if isinstance(v, dict) and "launches" in v}
            traffic_src = ("%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, KiB; SQ_VALU_MFMA_BUSY_CYCLES / "
                           "(1024 SIMDs x kernel duration x 2.4 GHz); separate passes, eager launches; build %s, git %s)"
                           % (tname, pj["build"].get("dv_version"), pj["build"].get("git_head")))
        fam, dom = roofline_families(live, pj, peak, flops_fwd / 1e9)
        d = fam[dom]
        rp = d["rocprofv3"]
        g = agg[dom]
        result["roofline"] = {
            # the family with the most kernel time in the forward (it was the GEMM kind by habit)
            "bound": "mfma", "family": dom, "kernel": d["kernel"], "time_share": d["time_share"],
            "achieved": d["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": d["frac"],
            "frac_rocprofv3": d["frac_rocprofv3"], "rocprofv3": None if rp is None else dict(rp, source=tname),
            "traffic": None if rp is None else rp["hbm_bytes_per_launch"], "traffic_source": traffic_src,
            "hbm_gbps": None if rp is None else rp["hbm_gbps"], "mfma_util": None if rp is None else rp["mfma_util"],
            "pmc_per_kernel_family": pmc_families,
            "flops_per_launch": d["gflop_per_forward"] * 1e9 / d["launches_per_forward"], "avg_launch_us": d["avg_launch_us"],
            "launches_per_forward": d["launches_per_forward"],
            "avg_op_us_event_pair_per_operation": 1e3 * g[2] / g[0], "achieved_event_pair_per_launch": g[1] / (g[2] * 1e-3) / 1e12,
            # every MFMA kernel family the same way; their FLOPs sum to forward.engine_counted_gflop
            "families": fam,
            "per_kind_ms_per_forward": {k: v[2] / reps for k, v in agg.items()},
            "per_kind_operations": {k: v[0] // reps for k, v in agg.items()},   # a split-K GEMM pair is one operation
            "forward": {"launches": n_launch,
                        # inside the sampler graph the time-embedding chain (4 launches) runs once per run, not per step
                        "launches_inside_sampler_loop": n_launch - (4 if (B <= 16 and os.environ.get("DVITS_TEMB_BATCH") != "0") else 0),
                        "algorithmic_gflop": flops_model(B, T, L) / 1e9,
                        "engine_counted_gflop": flops_fwd / 1e9, "ms_in_graph": fwd_ms,
                        "tflops": flops_model(B, T, L) / (fwd_ms * 1e-3) / 1e12,
                        "frac_of_peak": flops_model(B, T, L) / (fwd_ms * 1e-3) / 1e12 / peak},
        }
    if not args.no_roofline and world == 1:
        result["extra"] = lifecycle_extras(dev, args.precision)
        n_ho, bad = eng.handover_status()
        result["extra"]["in_epilogue_groupnorm_gemms"]["bench_shape"] = n_ho
        # (a timed-out hand-over is recovered from - fallback schedule, run repeated - but then this is not the default schedule's number)
        result["extra"]["in_epilogue_groupnorm_gemms"]["downgraded_to_separate_groupnorm"] = bool(eng.handover_downgraded)
        if bad or result["extra"]["in_epilogue_groupnorm_gemms"]["timed_out"] or eng.handover_downgraded:
            raise SystemExit("bench: an in-kernel GroupNorm hand-over timed out (is the GPU shared?) - not the default schedule's number")
    if not args.no_cpu_baseline and world == 1:        # reported at N = 1 only (rank 0's host cores, bounded sample)
        result["cpu_baseline"], y_ref = cpu_baseline(sd, B, T, L, S)
        result["speedup_vs_cpu_baseline"] = frames_per_s / result["cpu_baseline"]["value"] / world
        # the metric is frames/s PLUS the UNet-output error against the reference restatement -
        # one HIP forward on the inputs of the oracle forward the baseline leg just ran (same seed, t = PARITY_T)
        xo, co, eo, mo = (torch.from_numpy(a).to(dev) for a in synth.make_inputs(B, 80, T, L, seed=1234))
        with torch.no_grad():
            y = model(torch.cat([xo, co], 1), torch.full((B,), PARITY_T, device=dev), eo, encoder_attention_mask=mo).sample
        y = y.double().cpu()
        yr = y_ref.double()
        par = {"unet_rel_l2": float((y - yr).norm() / yr.norm()),
               "unet_max_abs_rel": float((y - yr).abs().max() / yr.abs().max()),
               "unet_parity": "one denoiser forward at the bench shape (B=%d, T=%d, L=%d, t=%.0f) on MI355X vs oracle/unet_ref.py "
                              "(torch-CPU fp32) on the same inputs; budget %.0e" % (B, T, L, PARITY_T, PARITY_TOL)}
        result.setdefault("extra", {}).update(par)
        if not (par["unet_rel_l2"] <= PARITY_TOL and par["unet_max_abs_rel"] <= PARITY_TOL):
            print(json.dumps(result))
            raise SystemExit("bench: UNet output differs from the oracle by %.3e (rel-L2) / %.3e (max-abs-rel) > %.0e"
                             % (par["unet_rel_l2"], par["unet_max_abs_rel"], PARITY_TOL))
    if world == 1 and not dry and not args.no_other_configs and not args.no_roofline and (B, T, L) == (8, 1024, 256):
        y_ref_ = locals().get("y_ref")
        # (the headline measurement and its parity result must survive whatever happens here: an out-of-memory on a smaller
        # or shared GPU, a hand-over time-out that check_or_recover turns into "repeat the run" - ADVICE r4)
        try:
            result.setdefault("extra", {}).update(other_configs(dev, model, sd, args.precision, S, y_ref_))
        except Exception as e:                       # noqa: BLE001
            result.setdefault("extra", {})["other_configs_error"] = "%s: %s" % (type(e).__name__, e)
    print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
