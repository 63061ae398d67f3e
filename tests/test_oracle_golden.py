"""CPU: the oracle (oracle/) against the golden vectors captured from the imported reference
(tools/make_golden.py).  This pins the oracle; the GPU tests then compare HIP against it."""
import numpy as np
import pytest
import torch

from conftest import UNET_CASES, oracle_cfg, rel_l2, unet_case
from oracle import sampler_ref, unet_ref


def _run_oracle(name, probes=None):
    kw, sd, sample, t, enc, mask = unet_case(name)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    ts = torch.from_numpy(t) if isinstance(t, np.ndarray) else t
    m = torch.from_numpy(mask[:, None, :].astype(np.float32)) if name == "durpred" else torch.from_numpy(mask)
    with torch.no_grad():
        return unet_ref.unet_forward(sdt, oracle_cfg(kw), torch.from_numpy(sample), ts, torch.from_numpy(enc), m,
                                     probes=probes)


@pytest.mark.parametrize("name", ["tiny", "oddT", "c100", "durpred", "cfg1"])
def test_unet_oracle_matches_reference_output(name, gold):
    y = _run_oracle(name).numpy()
    g = gold("unet_%s.npz" % name)
    assert y.shape == g["y"].shape
    assert rel_l2(y, g["y"]) < 1e-6     # same torch ops in the same order: rounding-level


def test_unet_oracle_probes(gold):
    probes = {}
    _run_oracle("tiny", probes)
    g = gold("unet_tiny.npz")
    for k, v in probes.items():
        v = v.numpy()
        if v.ndim == 3:
            assert rel_l2(v.mean(axis=2), g["probe_mean_" + k]) < 1e-5, k
            assert rel_l2(v[:, :, :8], g["probe_head_" + k]) < 1e-5, k
        else:
            assert rel_l2(v, g["probe_" + k]) < 1e-5, k


def test_fp32_noise_floor_recorded(gold):
    """The fp64 run stored with the golden shows where fp32 rounding sits (≈6e-7): the 1e-3
    budget is three orders of magnitude above it."""
    g = gold("unet_tiny.npz")
    assert rel_l2(g["y"], g["y64"]) < 5e-6


def _keys(g, prefix):
    return sorted(k[:-2] for k in g.files if k.startswith(prefix) and k.endswith("_x"))


def test_sampler_oracle_dpm(gold):
    g = gold("sampler_standin.npz")
    betas = torch.from_numpy(__import__("diff_vits_amd").synth.make_betas()) if False else None
    from diff_vits_amd import synth
    betas = torch.from_numpy(synth.make_betas())
    x = torch.from_numpy(g["x_sampler"])
    for key in _keys(g, "dpm_"):
        _, s, o, skip = key.split("_", 3)
        calls = []

        def model(xx, t_in):
            calls.append(float(t_in[0]))
            return sampler_ref.standin_model(xx, t_in)
        out, inter = sampler_ref.dpm_solver_pp_sample(model, betas, x.clone(), int(s[1:]), int(o[1:]), skip,
                                                      return_intermediate=True)
        assert rel_l2(out.numpy(), g[key + "_x"]) < 1e-6, key
        assert rel_l2(inter[0].numpy(), g[key + "_x1"]) < 1e-6, key
        assert np.allclose(np.array(calls, dtype=np.float32), g[key + "_tin"], rtol=0, atol=1e-3), key
        assert len(calls) == int(s[1:])            # NFE == steps


def test_sampler_oracle_unipc(gold):
    g = gold("sampler_standin.npz")
    from diff_vits_amd import synth
    betas = torch.from_numpy(synth.make_betas())
    x = torch.from_numpy(g["x_sampler"])[:1]
    for key in _keys(g, "unipc_"):
        _, s, o, variant = key.split("_", 3)
        out = sampler_ref.unipc_sample(sampler_ref.standin_model, betas, x.clone(), int(s[1:]), int(o[1:]),
                                       "time_uniform", variant)
        assert rel_l2(out.numpy(), g[key + "_x"]) < 1e-6, key


def test_schedule_known_answers(gold):
    g = gold("sampler_standin.npz")
    from diff_vits_amd import synth
    betas = torch.from_numpy(synth.make_betas())
    ns = sampler_ref.Schedule(betas, clip=True)
    t = torch.from_numpy(g["sched_t"])
    assert ns.total_N == int(g["sched_total_N"][0]) == 1000       # the -5.1 clip is a no-op for this schedule
    assert np.allclose(ns.log_alpha(t).numpy(), g["sched_dpm_log_alpha"], rtol=1e-6, atol=1e-7)
    assert np.allclose(ns.lam(t).numpy(), g["sched_dpm_lambda"], rtol=1e-5, atol=1e-5)
    assert np.allclose(ns.sigma(t).numpy(), g["sched_dpm_std"], rtol=1e-6, atol=1e-7)
    assert np.allclose(ns.inverse_lambda(torch.from_numpy(g["sched_inv_lambda_in"])).numpy(), g["sched_inv_lambda"],
                       rtol=1e-5, atol=1e-6)


@pytest.mark.slow
def test_sampler_oracle_real_unet_cfg1(gold):
    """BASELINE config 1 (B=1, C=80, T=256, L=128, 20-step DPM-Solver++ 2M) on the oracle UNet."""
    from diff_vits_amd import synth
    kw = UNET_CASES["cfg1"][0]
    _, sd, *_ = unet_case("cfg1")
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    x, cond, enc, mask = map(torch.from_numpy, synth.make_inputs(1, 80, 256, 128, seed=1234))
    betas = torch.from_numpy(synth.make_betas())
    with torch.no_grad():
        out = sampler_ref.dpm_solver_pp_sample(unet_ref.diffusion_model_fn(sdt, oracle_cfg(kw), cond, enc, mask), betas,
                                               x, 20, 2)
    assert rel_l2(out.numpy(), gold("sampler_cfg1.npz")["dpm_x"]) < 1e-5
