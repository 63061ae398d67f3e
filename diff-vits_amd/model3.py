"""Host-side mirror of the two reference classes that sit directly around the denoiser:

  model3.PromptEncoder      (reference model3.py:382-433)  -> native dv_penc_* (include/dvits_hip.h)
  model3.Diffusion_Encoder  (reference model3.py:867-914)  -> prompt encoder + channel concat + native UNet

Same class names, constructor keywords, parameter names/shapes (`load_state_dict` of a reference checkpoint works) and
call signatures.  `backend="hip"` (default; env DVITS_BACKEND) runs on libdvits_hip.so and raises without it or
without GPU tensors; `backend="torch"` is the explicit opt-in eager path (CPU / training), never a fallback.

The reference recomputes the prompt encoder in EVERY denoiser call although it does not depend on the step
(model3.py:906); Diffusion_Encoder here computes it (and the UNet's hoisted conditioning) once per
(prompt, prompt_lengths) pair and reuses it while the same tensors are passed again.
"""
import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from .unet1d.unet_1d_condition import UNet1DConditionModel


def sequence_mask(length, max_length=None):
    """reference commons.py:121-125."""
    if max_length is None:
        max_length = length.max()
    x = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


class ConvTBC(nn.Module):
    """reference model.py:137-151 (weight stored [k, C_in, C_out])."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0):
        super().__init__()
        self.kernel_size, self.padding = kernel_size, padding
        self.weight = nn.Parameter(torch.empty(kernel_size, in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x):
        return torch.conv_tbc(x.contiguous(), self.weight, self.bias, self.padding)


class ConvLayer(nn.Module):
    """reference model.py:153-171."""

    def __init__(self, c_in, c_out, kernel_size, dropout=0):
        super().__init__()
        self.layer_norm = nn.LayerNorm(c_in)
        self.conv = ConvTBC(c_in, c_out, kernel_size, padding=kernel_size // 2)
        nn.init.normal_(self.conv.weight, mean=0, std=math.sqrt((4 * (1.0 - dropout)) / (kernel_size * c_in)))

    def forward(self, x, encoder_padding_mask=None):
        if encoder_padding_mask is not None:
            x = x.masked_fill(encoder_padding_mask.t().unsqueeze(-1), 0)
        return self.conv(self.layer_norm(x))


class MultiheadAttention(nn.Module):
    """reference operations.py:304-416 as instantiated by EncSALayer: self-attention, fused in_proj_weight, no biases."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=False)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.xavier_uniform_(self.out_proj.weight)

    def forward(self, x, key_padding_mask):
        out, _ = F.multi_head_attention_forward(x, x, x, self.embed_dim, self.num_heads, self.in_proj_weight, None, None, None,
                                                False, 0.0, self.out_proj.weight, None, self.training, key_padding_mask, True,
                                                None)
        return out


class TransformerFFNLayer(nn.Module):
    """reference operations.py:644-693, padding 'SAME'."""

    def __init__(self, hidden_size, filter_size, kernel_size=9):
        super().__init__()
        self.kernel_size = kernel_size
        self.first_offset = -((kernel_size - 1) // 2)
        self.last_offset = self.first_offset + kernel_size - 1
        self.ffn_1 = nn.ModuleList([nn.Linear(hidden_size, filter_size, bias=(i == 0)) for i in range(kernel_size)])
        self.ffn_2 = nn.Linear(filter_size, hidden_size)

    def forward(self, x):
        padded = F.pad(x, (0, 0, 0, 0, -self.first_offset, self.last_offset))
        res = 0
        for i in range(self.kernel_size):
            shifted = padded[i:x.size(0) + i] if i else x      # tap 0 sees the unpadded x (reference :678)
            res = res + self.ffn_1[i](shifted)
        return self.ffn_2(F.relu(res * self.kernel_size ** -0.5))


class EncSALayer(nn.Module):
    """reference operations.py:784-821 (inference: dropout is the identity)."""

    def __init__(self, c, num_heads, kernel_size=9):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(c)
        self.self_attn = MultiheadAttention(c, num_heads)
        self.layer_norm2 = nn.LayerNorm(c)
        self.ffn = TransformerFFNLayer(c, 4 * c, kernel_size=kernel_size)

    def forward(self, x, encoder_padding_mask=None):
        keep = (1 - encoder_padding_mask.float()).transpose(0, 1)[..., None]
        x = (x + self.self_attn(self.layer_norm1(x), encoder_padding_mask)) * keep
        return (x + self.ffn(self.layer_norm2(x))) * keep


class TransformerEncoderLayer(nn.Module):
    """reference model.py:72-81 with layer == 8 (operations.py:961-964)."""

    def __init__(self, layer, hidden_size, dropout):
        super().__init__()
        if layer != 8:
            raise ValueError("only OPERATIONS_ENCODER[8] (EncSALayer, 8 heads, k=9) is used on this path")
        self.op = EncSALayer(hidden_size, 8, kernel_size=9)

    def forward(self, x, **kwargs):
        return self.op(x, **kwargs)


class PromptEncoder(nn.Module):
    """reference model3.py:382-433."""

    def __init__(self, in_channels=128, hidden_channels=512, out_channels=128, n_layers=6, p_dropout=0.2, last_ln=True,
                 gin_channels=None, backend=None):
        super().__init__()
        self.in_channels, self.hidden_size, self.out_channels = in_channels, hidden_channels, out_channels
        self.num_layers, self.dropout, self.last_ln = n_layers, p_dropout, last_ln
        self.layers = nn.ModuleList([TransformerEncoderLayer(8, hidden_channels, p_dropout) for _ in range(n_layers)])
        if last_ln:
            self.layer_norm = nn.LayerNorm(out_channels)
        self.pre = ConvLayer(in_channels, hidden_channels, 1, p_dropout)
        self.out_proj = ConvLayer(hidden_channels, out_channels, 1)
        if gin_channels is not None:
            self.g_proj = nn.Conv1d(gin_channels, in_channels, 1)
        self.backend = backend or os.environ.get("DVITS_BACKEND", "hip")
        if self.backend not in ("hip", "torch"):
            raise ValueError("backend must be 'hip' or 'torch', got %r" % (self.backend,))
        self._engine = None

    def hip_engine(self):
        if self._engine is None:
            from .engine import PromptEncoderEngine
            if not self.last_ln:
                raise ValueError("the native prompt encoder is built for last_ln=True (the reference call sites)")
            self._engine = PromptEncoderEngine(self)
        return self._engine

    def encode_channels_last(self, src_tokens, lengths, g=None):
        """[B, L, C_out] (the layout the denoiser's encoder_hidden_states takes), padding frames zero."""
        if g is not None:
            src_tokens = src_tokens + self.g_proj(g)
        keep = sequence_mask(lengths, src_tokens.size(2))
        if self.backend == "hip":
            if self.training and torch.is_grad_enabled():
                raise RuntimeError("backend='hip' is inference-only; construct with backend='torch' to train")
            return self.hip_engine().forward(src_tokens, keep.to(torch.float32))
        pad = ~keep
        keep_t = (1 - pad.float()).transpose(0, 1)[..., None]
        x = src_tokens.permute(2, 0, 1)
        x = self.pre(x, encoder_padding_mask=pad) * keep_t
        for layer in self.layers:
            x = layer(x, encoder_padding_mask=pad)
        x = self.out_proj(x) * keep_t
        if self.last_ln:
            x = self.layer_norm(x) * keep_t
        return x.permute(1, 0, 2)

    def forward(self, src_tokens, lengths, g=None):
        """src_tokens [B, C_in, L], lengths [B] -> [B, C_out, L] (reference layout; a view of the channels-last result)."""
        return self.encode_channels_last(src_tokens, lengths, g).transpose(1, 2)


class Diffusion_Encoder(nn.Module):
    """reference model3.py:867-914."""

    def __init__(self, in_channels=128, out_channels=128, hidden_channels=256, kernel_size=3, dilation_rate=2, n_layers=40,
                 n_heads=8, p_dropout=0.2, dim_time_mult=None, backend=None):
        super().__init__()
        self.in_channels, self.out_channels, self.hidden_channels = in_channels, out_channels, hidden_channels
        self.kernel_size, self.dilation_rate, self.n_layers, self.n_heads = kernel_size, dilation_rate, n_layers, n_heads
        self.unet = UNet1DConditionModel(
            in_channels=in_channels + hidden_channels, out_channels=out_channels, block_out_channels=(128, 256, 384, 512),
            norm_num_groups=8, cross_attention_dim=hidden_channels, attention_head_dim=n_heads, addition_embed_type="text",
            resnet_time_scale_shift="scale_shift", backend=backend)
        self.spec_channels = 513
        self.prompt_encoder = PromptEncoder(100, hidden_channels, hidden_channels, 4, 0.2, backend=backend)
        self.backend = self.unet.backend
        self._cond_key = None
        self._cond = None
        self._unet_cond_serial = None

    def _conditioning(self, prompt, prompt_lengths, dtype):
        """Step-invariant part of model3.py:904-906, 911: encoder output (re-masked), bool mask; cached while the SAME
        (prompt, prompt_lengths) tensor objects, unmodified, and the same weights are passed.  The cache entry holds the
        two tensors themselves: a key made of data_ptr()/_version alone can be met by a NEW prompt allocated at the freed
        address of the old one (the caching allocator recycles blocks), which then got the previous speaker's states."""
        wkey = tuple((p.data_ptr(), p._version) for p in self.prompt_encoder.parameters())
        key = self._cond_key
        hit = (key is not None and key[0] is prompt and key[1] == prompt._version and key[2] is prompt_lengths
               and key[3] == prompt_lengths._version and key[4] == wkey and key[5] == dtype)
        if not hit:
            mask = sequence_mask(prompt_lengths, prompt.size(2))
            enc = self.prompt_encoder.encode_channels_last(prompt, prompt_lengths) * mask.unsqueeze(-1).to(dtype)
            self._cond = (enc, mask.to(torch.bool))
            self._cond_key = (prompt, prompt._version, prompt_lengths, prompt_lengths._version, wkey, dtype)
            self._unet_cond_serial = None
        return self._cond

    def native_model(self, data):
        """The sampler-side handle for a whole run: the reference wraps `lambda x, t: diff_model(x, data, t)` into
        model_wrapper (model3.py:1173-1182); passing THIS object as `model` instead lets DPM_Solver / UniPC replay the
        complete loop as one hipGraph (prompt encoder evaluated once, here).  Also a plain callable (x, t_input)."""
        from .sampler._plan import NativeUNetModel
        cond, prompt, cond_lengths, prompt_lengths = data
        enc, mask = self._conditioning(prompt, prompt_lengths, torch.float32)
        return NativeUNetModel(self.unet, cond, enc, mask)

    def forward(self, x, data, t):
        cond, prompt, cond_lengths, prompt_lengths = data
        enc, mask = self._conditioning(prompt, prompt_lengths, x.dtype)
        if self.backend != "hip":
            assert torch.isnan(x).any() == False  # noqa: E712  (reference model3.py:903)
            return self.unet(torch.cat([x, cond], dim=1), t, enc, encoder_attention_mask=mask).sample
        # native path: the channel concat is a two-pointer read inside the engine, the conditioning (pooled-text
        # embedding, cross-attention K/V, mask bias) is set once per cached prompt; the per-step NaN assert of the
        # reference (a host sync per step) is not replayed
        unet = self.unet
        eng = unet.hip_engine()
        eng.sync_weights()
        B, cx, T = x.shape
        eng.prepare(B, T, enc.shape[1])
        # keyed on the engine's serials, not on the shape key: a re-plan of the SAME shape (the fused schedule retried after a
        # downgrade, a recovery after wait() == False) leaves the native handle unconditioned
        if self._unet_cond_serial != (eng.cond_serial, eng.prepare_serial):      # nobody else re-conditioned / re-planned meanwhile
            eng.set_cond(enc, unet._bias_from_mask(mask, torch.float32))
            self._unet_cond_serial = (eng.cond_serial, eng.prepare_serial)
        tt = unet._timesteps(t, x).detach().to(device=x.device, dtype=torch.float32).contiguous()
        xx, cc = x.detach().to(torch.float32).contiguous(), cond.detach().to(torch.float32).contiguous()
        y = eng.eval(xx, cc, tt)

        def again():         # (a timed-out in-launch hand-over, caught while this schedule's first result is verified: engine.py)
            eng.prepare(B, T, enc.shape[1])
            eng.set_cond(enc, unet._bias_from_mask(mask, torch.float32))
            self._unet_cond_serial = (eng.cond_serial, eng.prepare_serial)
            return eng.eval(xx, cc, tt)
        y2 = eng.result_leaves(again)       # no host synchronisation in the steady state (verify_handover_default)
        if y2 is not None:
            y = y2
        return y if x.dtype == torch.float32 else y.to(x.dtype)


def linear_beta_schedule(timesteps):
    """reference model3.py:935-942."""
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


class NaturalSpeech2(nn.Module):
    """Inference side of reference model3.NaturalSpeech2 (model3.py:955-1203): the diffusion model,
    the schedule buffers of __init__ (:976-1018, same names/values, so a reference checkpoint's buffers load) and
    `sample` for the 'unipc' and 'dpmsolver' methods.

    The VITS prior is not part of this package: pass any module whose `.infer(text, text_lengths, spec, spec_lengths,
    tone, language)` returns `(content [B, hidden, T], refer [B, 100, L])` as `vits=` (the reference's own `VITS`
    instance works: it is pure PyTorch), or call `sample_from_prior` with those two tensors.  Differences from the
    reference, all opt-in or fixes: `noise=` (explicit x_T; the reference draws torch.randn), batch > 1 for 'unipc'
    (the reference's wrapper only broadcasts at B = 1), and a 'dpmsolver' branch that runs (the
    reference's calls `vits.infer` with a tuple and cannot, quirk 7; here it uses the 'unipc' branch's plumbing with the
    solver call of model3.py:1148-1158).  On the HIP backend the whole solver loop is one hipGraph replay."""

    def __init__(self, cfg, vits=None, rvq_cross_entropy_loss_weight=0.1, diff_loss_weight=1.0, f0_loss_weight=1.0,
                 duration_loss_weight=1.0, ddim_sampling_eta=0, min_snr_loss_weight=False, min_snr_gamma=5, backend=None):
        super().__init__()
        if vits is not None:
            self.vits = vits
        self.diff_model = Diffusion_Encoder(**cfg["diffusion_encoder"], backend=backend)
        self.dim = self.diff_model.in_channels
        betas = linear_beta_schedule(cfg["train"]["timesteps"])
        alphas = 1. - betas
        alphas_cumprod = torch.cumprod(alphas, dim=0)
        alphas_cumprod_prev = F.pad(alphas_cumprod[:-1], (1, 0), value=1.)
        self.num_timesteps = betas.shape[0]
        self.sampling_timesteps = None
        self.ddim_sampling_eta = ddim_sampling_eta

        def register_buffer(name, val):
            self.register_buffer(name, val.to(torch.float32))

        register_buffer("betas", betas)
        register_buffer("alphas_cumprod", alphas_cumprod)
        register_buffer("alphas_cumprod_prev", alphas_cumprod_prev)
        register_buffer("sqrt_alphas_cumprod", torch.sqrt(alphas_cumprod))
        register_buffer("sqrt_one_minus_alphas_cumprod", torch.sqrt(1. - alphas_cumprod))
        register_buffer("log_one_minus_alphas_cumprod", torch.log(1. - alphas_cumprod))
        register_buffer("sqrt_recip_alphas_cumprod", torch.sqrt(1. / alphas_cumprod))
        register_buffer("sqrt_recipm1_alphas_cumprod", torch.sqrt(1. / alphas_cumprod - 1))
        posterior_variance = betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod)
        register_buffer("posterior_variance", posterior_variance)
        register_buffer("posterior_log_variance_clipped", torch.log(posterior_variance.clamp(min=1e-20)))
        register_buffer("posterior_mean_coef1", betas * torch.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod))
        register_buffer("posterior_mean_coef2", (1. - alphas_cumprod_prev) * torch.sqrt(alphas) / (1. - alphas_cumprod))
        snr = alphas_cumprod / (1 - alphas_cumprod)
        if min_snr_loss_weight:
            snr = snr.clamp(max=min_snr_gamma)
        register_buffer("loss_weight", snr)

    def sample_fun(self, x, t, data=None):
        """reference model3.py:1113-1118: the x_start prediction handed to model_wrapper."""
        return self.diff_model(x, data, t)

    @torch.no_grad()
    def sample_from_prior(self, content, refer, text_lengths, spec_lengths, vocos=None, sample_method="unipc", noise=None):
        """(content, refer) -> (audio | None, mel): reference model3.py:1162-1203 after the `vits.infer` call."""
        if sample_method not in ("unipc", "dpmsolver"):
            raise ValueError("sample_method %r is not supported (this build: 'unipc', 'dpmsolver')" % (sample_method,))
        shape = (content.shape[0], self.dim, content.shape[2])
        audio = torch.randn(shape, device=refer.device) if noise is None else noise.to(refer.device)
        if tuple(audio.shape) != shape:
            raise ValueError("noise must have shape %s, got %s" % (shape, tuple(audio.shape)))
        data = (content, refer, text_lengths, spec_lengths)
        native = self.diff_model.backend == "hip" and audio.is_cuda
        if sample_method == "unipc":
            from .sampler.uni_pc import NoiseScheduleVP, UniPC, model_wrapper
        else:
            from .sampler.dpm_solver import DPM_Solver, NoiseScheduleVP, model_wrapper
        if native:
            # schedule, solver (with its compiled plan and captured hipGraph) and the native model handle are kept per
            # sampling method: a new utterance only swaps the conditioning; same-shape utterances replay the graph
            cache = self.__dict__.setdefault("_native_samplers", {})
            bkey = (self.betas.data_ptr(), self.betas._version, str(self.betas.device))
            ent = cache.get(sample_method)
            if ent is None or ent["betas"] != bkey:
                noise_schedule = NoiseScheduleVP(schedule="discrete", betas=self.betas)
                nm = self.diff_model.native_model(data)
                model_fn = model_wrapper(nm, noise_schedule, model_type="x_start")
                solver = (UniPC(model_fn, noise_schedule, variant="bh2") if sample_method == "unipc"
                          else DPM_Solver(model_fn, noise_schedule, algorithm_type="dpmsolver++"))
                ent = cache[sample_method] = {"betas": bkey, "native": nm, "solver": solver}
            else:
                enc, mask = self.diff_model._conditioning(refer, spec_lengths, torch.float32)
                ent["native"].cond, ent["native"].enc, ent["native"].mask = content, enc, mask
            solver = ent["solver"]
        else:
            noise_schedule = NoiseScheduleVP(schedule="discrete", betas=self.betas)
            model_fn = model_wrapper(self.sample_fun, noise_schedule, model_type="x_start", model_kwargs={"data": data})
            solver = (UniPC(model_fn, noise_schedule, variant="bh2") if sample_method == "unipc"
                      else DPM_Solver(model_fn, noise_schedule, algorithm_type="dpmsolver++"))
        def run():
            if sample_method == "unipc":
                return solver.sample(audio, steps=30, order=2, skip_type="time_uniform", method="multistep")
            return solver.sample(audio, steps=40, order=2, skip_type="time_uniform", method="multistep")
        if native:
            # the mel leaves this package here (vocoder, caller): ONE host wait per utterance verifies the in-launch hand-overs
            # of the whole run (engine.UNetEngine.wait; the solver loop itself never blocks the host), and a lost run - a foreign
            # kernel shared the GPU - is repeated on the fallback schedule
            from .engine import HandoverLost
            eng = self.diff_model.unet.hip_engine()
            try:
                mel = run()
                ok = eng.unverified_results == 0 or eng.wait()
            except HandoverLost:                      # an EARLIER run's time-out, noticed by this call: the engine has recovered
                ok = False
            if not ok:
                mel = run()
                if not (eng.unverified_results == 0 or eng.wait()):
                    raise RuntimeError("in-kernel hand-over timed out on the fallback schedule (it has none): internal error")
        else:
            mel = run()
        if vocos is None:
            return None, mel
        vocos.to(mel.device)
        wav = vocos.decode(mel)
        if wav.ndim == 3:
            wav = wav.reshape(wav.shape[0], wav.shape[2])         # 'b 1 n -> b n'
        return wav, mel

    @torch.no_grad()
    def sample(self, text, spec, text_lengths, spec_lengths, tone, language, vocos, sampling_timesteps=200,
               sample_method="unipc", noise=None, prior_noise=None):
        """reference model3.py:1119-1203 (same positional signature).  `noise` = x_T, `prior_noise` = the prior's
        normal draw (forwarded as `noise=` to this package's VITS.infer); both default to fresh torch.randn draws."""
        self.sampling_timesteps = sampling_timesteps
        if not hasattr(self, "vits"):
            raise RuntimeError("NaturalSpeech2.sample needs the VITS prior: construct with vits=<module with .infer(...)> "
                               "or call sample_from_prior(content, refer, ...)")
        if prior_noise is None:
            content, refer = self.vits.infer(text, text_lengths, spec, spec_lengths, tone, language)
        else:
            content, refer = self.vits.infer(text, text_lengths, spec, spec_lengths, tone, language, noise=prior_noise)
        return self.sample_from_prior(content, refer, text_lengths, spec_lengths, vocos, sample_method, noise)


# ---- SURVEY.md §8f rank 3 (inference side of the VITS prior, from the text encoder's outputs onward) ------------------
def _conv1x1(conv, x, native):
    """k = 1 Conv1d on [B, C, T]: the implicit-GEMM kernel through the C ABI (dv_op_conv1d) on the HIP backend."""
    if not native:
        return conv(x)
    from . import _lib
    x32 = x.detach().to(torch.float32).contiguous()
    w = conv.weight.detach().to(torch.float32).contiguous()
    b = conv.bias.detach().to(torch.float32).contiguous()
    out = torch.empty((x32.shape[0], w.shape[0], x32.shape[2]), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().dv_op_conv1d(_lib.ptr(x32), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), x32.shape[0], x32.shape[1],
                                       x32.shape[2], w.shape[0], 1, 1, 0, _lib.PREC_BF16X3, _lib.stream_ptr()), "dv_op_conv1d")
    return out.to(x.dtype)


def generate_path(duration, mask):
    """reference commons.py:128-143.  duration [b, 1, t_x], mask [b, 1, t_y, t_x] -> monotonic alignment [b, 1, t_y, t_x]."""
    b, _, t_y, t_x = mask.shape
    cum = torch.cumsum(duration, -1).view(b * t_x)
    path = sequence_mask(cum, t_y).to(mask.dtype).view(b, t_x, t_y)
    path = path - F.pad(path, (0, 0, 1, 0, 0, 0))[:, :-1]
    return path.unsqueeze(1).transpose(2, 3) * mask


class DurationPredictor_unet(nn.Module):
    """reference model3.py:275-321: 1x1 convs around the UNet at the duration-predictor configuration
    (block_out_channels = (h/4, h/4, h/2, h/2), out_channels 1), integer timestep 1, float [B, 1, L] prompt mask."""

    def __init__(self, in_channels, hidden_channels, prompt_channels, kernel_size, p_dropout, out_channels=1, n_heads=8,
                 backend=None):
        super().__init__()
        self.n_heads, self.p_dropout = n_heads, p_dropout
        self.pre = nn.Conv1d(in_channels, hidden_channels, kernel_size=1)
        self.enc = UNet1DConditionModel(
            in_channels=in_channels, out_channels=out_channels,
            block_out_channels=(hidden_channels // 4, hidden_channels // 4, hidden_channels // 2, hidden_channels // 2),
            norm_num_groups=8, cross_attention_dim=hidden_channels, attention_head_dim=n_heads, addition_embed_type="text",
            resnet_time_scale_shift="scale_shift", backend=backend)
        self.prompt_proj = nn.Conv1d(prompt_channels, hidden_channels, 1)
        self.backend = self.enc.backend

    def forward(self, x, x_lengths, prompt, prompt_lengths):
        native = self.backend == "hip" and x.is_cuda
        x, prompt = x.detach(), prompt.detach()
        prompt = _conv1x1(self.prompt_proj, prompt, native)
        x_mask = torch.unsqueeze(sequence_mask(x_lengths, x.size(2)), 1).to(x.dtype)
        prompt_mask = torch.unsqueeze(sequence_mask(prompt_lengths, prompt.size(2)), 1).to(x.dtype)
        x = _conv1x1(self.pre, x, native) * x_mask
        prompt = prompt * prompt_mask
        x = self.enc(x, 1, prompt.transpose(1, 2), encoder_attention_mask=prompt_mask).sample
        return x * x_mask


class ChannelLayerNorm(nn.Module):
    """reference attentions.LayerNorm (attentions.py:12-24): LayerNorm over the channel axis of [B, C, T]; parameters
    `gamma` / `beta`."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x):
        return F.layer_norm(x.transpose(1, -1), (self.channels,), self.gamma, self.beta, self.eps).transpose(1, -1)


class RelativeMultiHeadAttention(nn.Module):
    """reference attentions.MultiHeadAttention (attentions.py:142-300) as the text encoder uses it: self-attention with
    windowed relative-position key / value embeddings shared by the heads.  The relative terms are written as band
    gathers (offset j - i within +-window) rather than the reference's pad-and-reshape skewing."""

    def __init__(self, channels, out_channels, n_heads, p_dropout=0.0, window_size=4):
        super().__init__()
        assert channels % n_heads == 0
        self.n_heads, self.window_size, self.k_channels = n_heads, window_size, channels // n_heads
        self.conv_q = nn.Conv1d(channels, channels, 1)
        self.conv_k = nn.Conv1d(channels, channels, 1)
        self.conv_v = nn.Conv1d(channels, channels, 1)
        self.conv_o = nn.Conv1d(channels, out_channels, 1)
        std = self.k_channels ** -0.5
        self.emb_rel_k = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * std)
        self.emb_rel_v = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * std)

    def forward(self, x, c, attn_mask=None):
        b, ch, t = x.shape
        h, d, w = self.n_heads, self.k_channels, self.window_size
        q = self.conv_q(x).view(b, h, d, t).transpose(2, 3) / math.sqrt(d)
        k = self.conv_k(c).view(b, h, d, t).transpose(2, 3)
        v = self.conv_v(c).view(b, h, d, t).transpose(2, 3)
        i = torch.arange(t, device=x.device).unsqueeze(1)
        j = torch.arange(t, device=x.device).unsqueeze(0)
        off = j - i + w
        band = (off >= 0) & (off <= 2 * w)
        scores = torch.matmul(q, k.transpose(-2, -1))
        q_rel = torch.matmul(q, self.emb_rel_k[0].t())
        scores = scores + q_rel.gather(-1, off.clamp(0, 2 * w).expand(b, h, t, t)) * band
        if attn_mask is not None:
            scores = scores.masked_fill(attn_mask == 0, -1e4)
        p_attn = F.softmax(scores, dim=-1)
        out = torch.matmul(p_attn, v)
        jj = i + torch.arange(2 * w + 1, device=x.device).unsqueeze(0) - w
        valid = (jj >= 0) & (jj < t)
        rel_w = p_attn.gather(-1, jj.clamp(0, t - 1).expand(b, h, t, 2 * w + 1)) * valid
        out = out + torch.matmul(rel_w, self.emb_rel_v[0])
        return self.conv_o(out.transpose(2, 3).contiguous().view(b, ch, t))


class FFN(nn.Module):
    """reference attentions.FFN (attentions.py:322-380): two 'same'-padded convs with ReLU, masked."""

    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0.0):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv_1 = nn.Conv1d(in_channels, filter_channels, kernel_size)
        self.conv_2 = nn.Conv1d(filter_channels, out_channels, kernel_size)

    def forward(self, x, x_mask):
        pad = ((self.kernel_size - 1) // 2, self.kernel_size // 2)
        x = torch.relu(self.conv_1(F.pad(x * x_mask, pad)))
        return self.conv_2(F.pad(x * x_mask, pad)) * x_mask


class Encoder(nn.Module):
    """reference attentions.Encoder (attentions.py:37-88)."""

    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0.0, window_size=4,
                 gin_channels=0, cond_layer_idx=2):
        super().__init__()
        self.n_layers = n_layers
        self.cond_layer_idx = n_layers
        if gin_channels != 0:
            self.spk_emb_linear = nn.Linear(gin_channels, hidden_channels)
            self.cond_layer_idx = cond_layer_idx
            assert self.cond_layer_idx < n_layers, "cond_layer_idx should be less than n_layers"
        self.attn_layers = nn.ModuleList([RelativeMultiHeadAttention(hidden_channels, hidden_channels, n_heads, p_dropout, window_size)
                                          for _ in range(n_layers)])
        self.norm_layers_1 = nn.ModuleList([ChannelLayerNorm(hidden_channels) for _ in range(n_layers)])
        self.ffn_layers = nn.ModuleList([FFN(hidden_channels, hidden_channels, filter_channels, kernel_size, p_dropout)
                                         for _ in range(n_layers)])
        self.norm_layers_2 = nn.ModuleList([ChannelLayerNorm(hidden_channels) for _ in range(n_layers)])

    def forward(self, x, x_mask, g=None):
        attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
        x = x * x_mask
        for i in range(self.n_layers):
            if i == self.cond_layer_idx and g is not None:
                x = (x + self.spk_emb_linear(g.transpose(1, 2)).transpose(1, 2)) * x_mask
            x = self.norm_layers_1[i](x + self.attn_layers[i](x, x, attn_mask))
            x = self.norm_layers_2[i](x + self.ffn_layers[i](x, x_mask))
        return x * x_mask


class TextEncoder(nn.Module):
    """reference model3.TextEncoder (model3.py:322-381).  The vocabulary sizes come from the reference's `text`
    package there (len(symbols), num_tones, num_languages = 108 / 11 / 3 in this checkout); here they are arguments.
    Once per utterance over <= a few hundred tokens: plain torch ops on the caller's device."""

    def __init__(self, n_vocab, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout,
                 gin_channels=0, n_tones=11, n_languages=3):
        super().__init__()
        self.out_channels, self.hidden_channels = out_channels, hidden_channels
        self.emb = nn.Embedding(n_vocab, hidden_channels)
        self.tone_emb = nn.Embedding(n_tones, hidden_channels)
        self.language_emb = nn.Embedding(n_languages, hidden_channels)
        for e in (self.emb, self.tone_emb, self.language_emb):
            nn.init.normal_(e.weight, 0.0, hidden_channels ** -0.5)
        self.encoder = Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout, gin_channels=gin_channels)
        self.proj = nn.Conv1d(hidden_channels, out_channels * 2, 1)

    def forward(self, x, x_lengths, tone, language, g=None):
        x = (self.emb(x) + self.tone_emb(tone) + self.language_emb(language)) * math.sqrt(self.hidden_channels)
        x = torch.transpose(x, 1, -1)
        x_mask = torch.unsqueeze(sequence_mask(x_lengths, x.size(2)), 1).to(x.dtype)
        x = self.encoder(x * x_mask, x_mask, g=g)
        stats = self.proj(x) * x_mask
        m, logs = torch.split(stats, self.out_channels, dim=1)
        return x, m, logs, x_mask


class VITS(nn.Module):
    """Inference side of the reference's prior (model3.py:646-860): `infer` = reference encoder (TextTimeEmbedding, one
    head) -> text encoder -> DurationPredictor_unet -> monotonic alignment -> prior sample -> 6-layer `o_proj`
    PromptEncoder with speaker conditioning.  Parameter names of `ref_enc.*`, `dp.*`, `o_proj.*` are the reference's.

    With `n_vocab` given the text encoder `enc_p` (TextEncoder over the relative-position Encoder, reference parameter
    names) is built too; otherwise pass any module with the reference signature
    `enc_p(x, x_lengths, tone, language, g) -> (x, m_p, logs_p, x_mask)`, or call `infer_from_encoder`.  The duration
    predictor's UNet and `o_proj` run on the HIP engine (backend='hip'); the once-per-utterance glue (reference encoder,
    alignment path, gathers) is a handful of torch ops on the same device.  `noise=` fixes the prior noise (the reference
    draws torch.randn_like)."""

    def __init__(self, n_vocab=None, spec_channels=None, inter_channels=128, hidden_channels=256, filter_channels=256,
                 n_heads=2, n_layers=6, kernel_size=3, p_dropout=0.1, gin_channels=256, enc_p=None, n_tones=11, n_languages=3,
                 backend=None, **unused):
        super().__init__()
        from .unet1d.embeddings import TextTimeEmbedding
        self.inter_channels, self.hidden_channels, self.gin_channels = inter_channels, hidden_channels, gin_channels
        if enc_p is not None:
            self.enc_p = enc_p
        elif n_vocab is not None:
            self.enc_p = TextEncoder(n_vocab, inter_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size,
                                     p_dropout, gin_channels, n_tones, n_languages)
        self.dp = DurationPredictor_unet(hidden_channels, 256, 100, 3, 0.5, backend=backend)
        self.ref_enc = TextTimeEmbedding(100, gin_channels, 1)
        self.o_proj = PromptEncoder(inter_channels, hidden_channels, inter_channels, 6, 0.2, gin_channels=gin_channels,
                                    backend=backend)

    @torch.no_grad()
    def infer_from_encoder(self, x, m_p, logs_p, x_mask, x_lengths, y, y_lengths, g=None, noise_scale=0.667, length_scale=1,
                           noise=None):
        """reference model3.py:839-860 (after the `enc_p` call).  Returns (z [B, inter, T'], y, y_lengths')."""
        if g is None:
            g = self.ref_enc(y.transpose(1, 2)).unsqueeze(-1)
        logw = self.dp(x, x_lengths, y, y_lengths)
        w = torch.exp(logw) * x_mask * length_scale
        w_ceil = torch.ceil(w)
        y_len = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
        y_mask = torch.unsqueeze(sequence_mask(y_len, None), 1).to(x_mask.dtype)
        attn = generate_path(w_ceil, torch.unsqueeze(x_mask, 2) * torch.unsqueeze(y_mask, -1))
        m_p = torch.matmul(attn.squeeze(1), m_p.transpose(1, 2)).transpose(1, 2)
        logs_p = torch.matmul(attn.squeeze(1), logs_p.transpose(1, 2)).transpose(1, 2)
        eps = torch.randn_like(m_p) if noise is None else noise.to(m_p)
        z_p = m_p + eps * torch.exp(logs_p) * noise_scale
        return self.o_proj(z_p, y_len, g), y, y_len

    @torch.no_grad()
    def infer(self, x, x_lengths, y, y_lengths, tone, language, noise_scale=0.667, length_scale=1, noise_scale_w=0.8,
              max_len=None, sdp_ratio=0, noise=None):
        """reference model3.py:817-860 (same positional signature); returns (z, y)."""
        if not hasattr(self, "enc_p"):
            raise RuntimeError("VITS.infer needs the text encoder: construct with enc_p=<module> or call infer_from_encoder")
        g = self.ref_enc(y.transpose(1, 2)).unsqueeze(-1)
        x, m_p, logs_p, x_mask = self.enc_p(x, x_lengths, tone, language, g)
        z, y, _ = self.infer_from_encoder(x, m_p, logs_p, x_mask, x_lengths, y, y_lengths, g, noise_scale, length_scale, noise)
        return z, y
