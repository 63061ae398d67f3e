"""`UNet1DConditionModel`: host-side mirror of the reference denoiser interface
(reference unet1d/unet_1d_condition.py:61-1037) over the MI355X HIP engine.

Same constructor keywords, parameter names/shapes (so `load_state_dict` of a reference
checkpoint works), `forward` signature and `UNet1DConditionOutput` return type as the
reference.  `backend="hip"` (default) runs the forward through libdvits_hip.so on the
current HIP stream and raises if the library or a GPU tensor is missing; `backend="torch"`
(explicit opt-in: constructor kwarg or env DVITS_BACKEND=torch) runs the eager torch
modules of this package (training, CPU use).  There is no silent fallback between the two.
"""
import os
from collections import OrderedDict
from typing import Optional, Tuple, Union

import torch
from torch import nn

from .blocks import DownStage, MidStage, UpStage
from .embeddings import TextTimeEmbedding, TimestepEmbedding, Timesteps


class UNet1DConditionOutput(OrderedDict):
    """Return container of `UNet1DConditionModel.forward` (reference
    unet_1d_condition.py:48-58 / outputs.py:36-104): attribute, key and index access to
    `.sample` [B, out_channels, T]."""

    def __init__(self, sample=None):
        super().__init__()
        self["sample"] = sample

    @property
    def sample(self):
        return self["sample"]

    def __getitem__(self, k):
        if isinstance(k, str):
            return super().__getitem__(k)
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


_SUPPORTED_DOWN = ("CrossAttnDownBlock2D", "DownBlock2D")
_SUPPORTED_UP = ("UpBlock2D", "CrossAttnUpBlock2D")


class UNet1DConditionModel(nn.Module):
    """Conditional 1-D UNet: timestep + pooled-text embedding, cross-attention on
    `encoder_hidden_states`.  See module docstring for the backend contract."""

    def __init__(
        self,
        sample_size: Optional[int] = None,
        in_channels: int = 4,
        out_channels: int = 4,
        center_input_sample: bool = False,
        flip_sin_to_cos: bool = True,
        freq_shift: int = 0,
        down_block_types: Tuple[str] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                        "DownBlock2D"),
        mid_block_type: Optional[str] = "UNetMidBlock2DCrossAttn",
        up_block_types: Tuple[str] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D",
                                      "CrossAttnUpBlock2D"),
        only_cross_attention=False,
        block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
        layers_per_block: int = 2,
        downsample_padding: int = 1,
        mid_block_scale_factor: float = 1,
        act_fn: str = "silu",
        norm_num_groups: Optional[int] = 32,
        norm_eps: float = 1e-5,
        cross_attention_dim: int = 1280,
        transformer_layers_per_block: int = 1,
        encoder_hid_dim: Optional[int] = None,
        encoder_hid_dim_type: Optional[str] = None,
        attention_head_dim: int = 8,
        num_attention_heads: Optional[int] = None,
        dual_cross_attention: bool = False,
        use_linear_projection: bool = False,
        class_embed_type: Optional[str] = None,
        addition_embed_type: Optional[str] = None,
        addition_time_embed_dim: Optional[int] = None,
        num_class_embeds: Optional[int] = None,
        upcast_attention: bool = False,
        resnet_time_scale_shift: str = "default",
        resnet_skip_time_act: bool = False,
        resnet_out_scale_factor: int = 1.0,
        time_embedding_type: str = "positional",
        time_embedding_dim: Optional[int] = None,
        time_embedding_act_fn: Optional[str] = None,
        timestep_post_act: Optional[str] = None,
        time_cond_proj_dim: Optional[int] = None,
        conv_in_kernel: int = 3,
        conv_out_kernel: int = 3,
        projection_class_embeddings_input_dim: Optional[int] = None,
        class_embeddings_concat: bool = False,
        mid_block_only_cross_attention: Optional[bool] = None,
        cross_attention_norm: Optional[str] = None,
        addition_embed_type_num_heads=64,
        backend: Optional[str] = None,
    ):
        super().__init__()
        if num_attention_heads is not None:
            raise ValueError("`num_attention_heads` cannot be passed; the head count is `attention_head_dim` "
                             "(reference unet_1d_condition.py:208-219).")
        if len(down_block_types) != len(up_block_types) or len(block_out_channels) != len(down_block_types):
            raise ValueError("`down_block_types`, `up_block_types` and `block_out_channels` must have equal length.")
        unsupported = []
        if any(t not in _SUPPORTED_DOWN for t in down_block_types) or any(t not in _SUPPORTED_UP for t in up_block_types):
            unsupported.append("block types other than CrossAttnDown/Down/Up/CrossAttnUp")
        if mid_block_type != "UNetMidBlock2DCrossAttn":
            unsupported.append("mid_block_type=%r" % (mid_block_type,))
        if resnet_time_scale_shift != "scale_shift":
            unsupported.append("resnet_time_scale_shift=%r" % (resnet_time_scale_shift,))
        if addition_embed_type != "text":
            unsupported.append("addition_embed_type=%r" % (addition_embed_type,))
        for name, val, want in (
            ("center_input_sample", center_input_sample, False), ("only_cross_attention", only_cross_attention, False),
            ("act_fn", act_fn, "silu"), ("encoder_hid_dim", encoder_hid_dim, None),
            ("encoder_hid_dim_type", encoder_hid_dim_type, None), ("dual_cross_attention", dual_cross_attention, False),
            ("use_linear_projection", use_linear_projection, False), ("class_embed_type", class_embed_type, None),
            ("num_class_embeds", num_class_embeds, None), ("resnet_skip_time_act", resnet_skip_time_act, False),
            ("time_embedding_type", time_embedding_type, "positional"), ("time_embedding_act_fn", time_embedding_act_fn, None),
            ("timestep_post_act", timestep_post_act, None), ("time_cond_proj_dim", time_cond_proj_dim, None),
            ("conv_in_kernel", conv_in_kernel, 3), ("conv_out_kernel", conv_out_kernel, 3),
            ("class_embeddings_concat", class_embeddings_concat, False), ("cross_attention_norm", cross_attention_norm, None),
            ("transformer_layers_per_block", transformer_layers_per_block, 1), ("downsample_padding", downsample_padding, 1),
            ("mid_block_scale_factor", mid_block_scale_factor, 1), ("resnet_out_scale_factor", resnet_out_scale_factor, 1.0),
        ):
            if val != want:
                unsupported.append("%s=%r" % (name, val))
        if not isinstance(layers_per_block, int) or not isinstance(cross_attention_dim, int) \
                or not isinstance(attention_head_dim, int) or norm_num_groups is None:
            unsupported.append("per-block tuples for layers_per_block/cross_attention_dim/attention_head_dim, or no GroupNorm")
        if unsupported:
            raise ValueError("UNet1DConditionModel (MI355X build) supports the diffusion-sampler configuration family "
                             "only; unsupported: " + "; ".join(unsupported))
        for i, t in enumerate(down_block_types):
            if (t == "DownBlock2D") != (i == len(down_block_types) - 1):
                raise ValueError("supported layout is CrossAttnDownBlock2D x(n-1) + DownBlock2D")
        for i, t in enumerate(up_block_types):
            if (t == "UpBlock2D") != (i == 0):
                raise ValueError("supported layout is UpBlock2D + CrossAttnUpBlock2D x(n-1)")

        self.sample_size = sample_size
        heads = attention_head_dim
        chans = tuple(block_out_channels)
        time_embed_dim = time_embedding_dim or chans[0] * 4

        self.conv_in = nn.Conv1d(in_channels, chans[0], 3, padding=1)
        self.time_proj = Timesteps(chans[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(chans[0], time_embed_dim)
        self.add_embedding = TextTimeEmbedding(cross_attention_dim, time_embed_dim, num_heads=addition_embed_type_num_heads)

        self.down_blocks = nn.ModuleList()
        out_ch = chans[0]
        n = len(chans)
        for i in range(n):
            in_ch, out_ch = out_ch, chans[i]
            last = i == n - 1
            self.down_blocks.append(DownStage(in_ch, out_ch, time_embed_dim, layers_per_block, norm_num_groups,
                                              norm_eps, heads, cross_attention_dim, with_attention=not last,
                                              add_downsample=not last))
        self.mid_block = MidStage(chans[-1], time_embed_dim, norm_num_groups, norm_eps, heads, cross_attention_dim)

        self.up_blocks = nn.ModuleList()
        rev = list(reversed(chans))
        out_ch = rev[0]
        self.num_upsamplers = 0
        for i in range(n):
            prev_out, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, n - 1)]
            last = i == n - 1
            self.num_upsamplers += 0 if last else 1
            self.up_blocks.append(UpStage(in_ch, out_ch, prev_out, time_embed_dim, layers_per_block + 1,
                                          norm_num_groups, norm_eps, heads, cross_attention_dim,
                                          with_attention=i > 0, add_upsample=not last))

        self.conv_norm_out = nn.GroupNorm(norm_num_groups, chans[0], eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv1d(chans[0], out_channels, 3, padding=1)

        ctor = dict(locals())
        self.config = {k: ctor[k] for k in (
            "sample_size", "in_channels", "out_channels", "center_input_sample", "flip_sin_to_cos", "freq_shift",
            "down_block_types", "mid_block_type", "up_block_types", "only_cross_attention", "block_out_channels",
            "layers_per_block", "downsample_padding", "mid_block_scale_factor", "act_fn", "norm_num_groups",
            "norm_eps", "cross_attention_dim", "transformer_layers_per_block", "encoder_hid_dim",
            "encoder_hid_dim_type", "attention_head_dim", "num_attention_heads", "dual_cross_attention",
            "use_linear_projection", "class_embed_type", "addition_embed_type", "addition_time_embed_dim",
            "num_class_embeds", "upcast_attention", "resnet_time_scale_shift", "resnet_skip_time_act",
            "resnet_out_scale_factor", "time_embedding_type", "time_embedding_dim", "time_embedding_act_fn",
            "timestep_post_act", "time_cond_proj_dim", "conv_in_kernel", "conv_out_kernel",
            "projection_class_embeddings_input_dim", "class_embeddings_concat", "mid_block_only_cross_attention",
            "cross_attention_norm", "addition_embed_type_num_heads")}
        self.config["num_attention_heads"] = heads

        self.backend = backend or os.environ.get("DVITS_BACKEND", "hip")
        if self.backend not in ("hip", "torch"):
            raise ValueError("backend must be 'hip' or 'torch', got %r" % (self.backend,))
        if self.backend == "hip":
            # the native timestep-embedding kernel (csrc/kernels_misc.hip k_timestep_sincos) is the diffusion
            # configuration's: [cos | sin], no frequency shift, E = 4 * block_out_channels[0].  The torch mirror honours
            # other values; the HIP backend must refuse them rather than compute something else silently.
            hip_only = [("flip_sin_to_cos", flip_sin_to_cos, True), ("freq_shift", freq_shift, 0),
                        ("time_embedding_dim", time_embedding_dim, None)]
            bad = ["%s=%r" % (k, v) for k, v, want in hip_only if v != want and not (k == "time_embedding_dim" and v == chans[0] * 4)]
            if bad:
                raise ValueError("UNet1DConditionModel backend='hip' does not implement " + "; ".join(bad)
                                 + " (construct with backend='torch' for these)")
        self._engine = None

    # ------------------------------------------------------------------ engine plumbing
    def hip_engine(self, precision: Optional[str] = None):
        """The native engine bound to this module's current parameters (created and packed on
        first use; re-packed when parameters changed, e.g. after load_state_dict/.to())."""
        import importlib
        engine_mod = importlib.import_module("diff_vits_amd.engine")
        if self._engine is None:
            self._engine = engine_mod.UNetEngine(self)
        self._engine.sync_weights(precision)
        return self._engine

    # ------------------------------------------------------------------ forward
    def _bias_from_mask(self, mask, dtype):
        # reference unet_1d_condition.py:816-818
        if mask is None:
            return None
        bias = (1 - mask.to(dtype)) * -10000.0
        return bias.unsqueeze(1) if bias.dim() == 2 else bias

    def _timesteps(self, timestep, sample):
        # reference unet_1d_condition.py:825-839
        if not torch.is_tensor(timestep):
            dtype = torch.float64 if isinstance(timestep, float) else torch.int64
            timestep = torch.tensor([timestep], dtype=dtype, device=sample.device)
        elif timestep.dim() == 0:
            timestep = timestep[None].to(sample.device)
        return timestep.expand(sample.shape[0])

    def forward(
        self,
        sample: torch.Tensor,
        timestep: Union[torch.Tensor, float, int],
        encoder_hidden_states: torch.Tensor,
        class_labels=None,
        timestep_cond=None,
        attention_mask=None,
        cross_attention_kwargs=None,
        added_cond_kwargs=None,
        down_block_additional_residuals=None,
        mid_block_additional_residual=None,
        encoder_attention_mask: Optional[torch.Tensor] = None,
        return_dict: bool = True,
    ):
        for name, val in (("class_labels", class_labels), ("timestep_cond", timestep_cond),
                          ("attention_mask", attention_mask), ("cross_attention_kwargs", cross_attention_kwargs),
                          ("added_cond_kwargs", added_cond_kwargs),
                          ("down_block_additional_residuals", down_block_additional_residuals),
                          ("mid_block_additional_residual", mid_block_additional_residual)):
            if val is not None:
                raise ValueError("UNet1DConditionModel.forward: `%s` is not supported on the sampling path" % name)

        timesteps = self._timesteps(timestep, sample)
        bias = self._bias_from_mask(encoder_attention_mask, sample.dtype)

        if self.backend == "hip":
            if self.training and torch.is_grad_enabled():
                raise RuntimeError("backend='hip' is inference-only; construct with backend='torch' to train")
            out = self.hip_engine().forward(sample, timesteps, encoder_hidden_states, bias, mask_src=encoder_attention_mask)
        else:
            out = self._forward_torch(sample, timesteps, encoder_hidden_states, bias)
        if not return_dict:
            return (out,)
        return UNet1DConditionOutput(sample=out)

    def _forward_torch(self, sample, timesteps, enc, bias):
        """Eager torch forward (reference unet_1d_condition.py:785-1037 restricted to the
        supported configuration)."""
        up_factor = 2 ** self.num_upsamplers
        forward_upsample_size = any(s % up_factor != 0 for s in sample.shape[-2:])
        emb = self.time_embedding(self.time_proj(timesteps).to(sample.dtype)) + self.add_embedding(enc)
        h = self.conv_in(sample)
        skips = (h,)
        for blk in self.down_blocks:
            h, outs = blk(h, emb, enc, bias)
            skips += outs
        h = self.mid_block(h, emb, enc, bias)
        for i, blk in enumerate(self.up_blocks):
            n_res = len(blk.resnets)
            res, skips = skips[-n_res:], skips[:-n_res]
            size = None
            if i != len(self.up_blocks) - 1 and forward_upsample_size:
                size = skips[-1].shape[2:]
            h = blk(h, res, emb, enc, bias, size)
        return self.conv_out(self.conv_act(self.conv_norm_out(h)))
