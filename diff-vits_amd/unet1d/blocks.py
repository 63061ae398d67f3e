"""Building blocks of the 1-D conditional UNet: parameter containers whose attribute
names reproduce the reference state-dict layout (reference unet1d/resnet.py,
unet1d/attention.py, unet1d/attention_processor.py, unet1d/transformer_1d.py,
unet1d/unet_1d_blocks.py), plus an eager torch forward for each (the
`backend="torch"` path).  Only the block types the diffusion sampler instantiates
are provided; on a GPU the HIP engine reads the
parameters of these containers and never calls these forwards.
"""
import torch
import torch.nn.functional as F
from torch import nn


class ResnetBlock1D(nn.Module):
    """GN -> SiLU -> conv3 -> (temb scale/shift on GN) -> SiLU -> conv3 -> + shortcut
    (reference ResnetBlock2D, unet1d/resnet.py:461-641, time_embedding_norm='scale_shift')."""

    def __init__(self, in_channels, out_channels, temb_channels, groups, eps):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv1d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, 2 * out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv1d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv1d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        scale, shift = self.time_emb_proj(F.silu(temb))[:, :, None].chunk(2, dim=1)
        h = self.norm2(h) * (1 + scale) + shift
        h = self.conv2(F.silu(h))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample1D(nn.Module):
    """Conv1d(k3, s2, p1) (reference Downsample2D, unet1d/resnet.py:176-223)."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv1d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample1D(nn.Module):
    """Nearest x2 (or to an explicit size) then Conv1d(k3, p1) (reference Upsample2D,
    unet1d/resnet.py:104-173)."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv1d(channels, channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


class Attention(nn.Module):
    """Multi-head attention with bias-free q/k/v and a biased output projection
    (reference Attention + AttnProcessor2_0, unet1d/attention_processor.py:26-154, 971-1052)."""

    def __init__(self, query_dim, cross_attention_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv_dim, inner, bias=False)
        self.to_v = nn.Linear(kv_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])

    def forward(self, x, context=None, bias=None):
        B, H = x.shape[0], self.heads
        src = x if context is None else context
        q = self.to_q(x).view(B, -1, H, self.to_q.out_features // H).transpose(1, 2)
        k = self.to_k(src).view(B, -1, H, q.shape[-1]).transpose(1, 2)
        v = self.to_v(src).view(B, -1, H, q.shape[-1]).transpose(1, 2)
        mask = None if bias is None else bias[:, None].expand(B, H, 1, bias.shape[-1])
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        return self.to_out[0](o.transpose(1, 2).reshape(B, -1, H * q.shape[-1]))


class GEGLU(nn.Module):
    """proj -> a * gelu_erf(gate) (reference unet1d/attention.py:280-301)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        a, gate = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(gate)


class FeedForward(nn.Module):
    """net = [GEGLU, Dropout, Linear] (reference unet1d/attention.py:206-255)."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """LN -> self-attn, LN -> cross-attn, LN -> GEGLU FF, each with a residual
    (reference unet1d/attention.py:26-203)."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context, bias):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context, bias) + x
        return self.ff(self.norm3(x)) + x


class Transformer1DModel(nn.Module):
    """GN(eps 1e-6) -> 1x1 conv -> transformer block on (B,T,C) -> 1x1 conv -> + residual
    (reference Transformer2DModel, unet1d/transformer_1d.py:41-326, continuous input)."""

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Conv1d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Conv1d(inner, in_channels, 1)

    def forward(self, x, context, bias):
        h = self.proj_in(self.norm(x)).permute(0, 2, 1)
        for blk in self.transformer_blocks:
            h = blk(h, context, bias)
        return self.proj_out(h.permute(0, 2, 1).contiguous()) + x


class DownStage(nn.Module):
    """`CrossAttnDownBlock2D` (with attention) or `DownBlock2D` (without)
    (reference unet1d/unet_1d_blocks.py:861-1097)."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, groups, eps, heads,
                 cross_attention_dim, with_attention, add_downsample):
        super().__init__()
        self.has_cross_attention = with_attention
        self.resnets = nn.ModuleList([
            ResnetBlock1D(in_channels if i == 0 else out_channels, out_channels, temb_channels, groups, eps)
            for i in range(num_layers)])
        if with_attention:
            self.attentions = nn.ModuleList([
                Transformer1DModel(heads, out_channels // heads, out_channels, cross_attention_dim, groups)
                for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample1D(out_channels)]) if add_downsample else None

    def forward(self, h, temb, context=None, bias=None):
        outs = ()
        for i, res in enumerate(self.resnets):
            h = res(h, temb)
            if self.has_cross_attention:
                h = self.attentions[i](h, context, bias)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class MidStage(nn.Module):
    """`UNetMidBlock2DCrossAttn` (reference unet1d/unet_1d_blocks.py:516-623)."""

    def __init__(self, channels, temb_channels, groups, eps, heads, cross_attention_dim):
        super().__init__()
        self.has_cross_attention = True
        self.attentions = nn.ModuleList([
            Transformer1DModel(heads, channels // heads, channels, cross_attention_dim, groups)])
        self.resnets = nn.ModuleList([
            ResnetBlock1D(channels, channels, temb_channels, groups, eps) for _ in range(2)])

    def forward(self, h, temb, context=None, bias=None):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, context, bias)
        return self.resnets[1](h, temb)


class UpStage(nn.Module):
    """`CrossAttnUpBlock2D` (with attention) or `UpBlock2D` (without)
    (reference unet1d/unet_1d_blocks.py:1986-2207).  Each resnet consumes
    cat([hidden, skip], channel)."""

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, groups, eps,
                 heads, cross_attention_dim, with_attention, add_upsample):
        super().__init__()
        self.has_cross_attention = with_attention
        resnets = []
        for i in range(num_layers):
            skip_ch = in_channels if i == num_layers - 1 else out_channels
            res_in = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock1D(res_in + skip_ch, out_channels, temb_channels, groups, eps))
        self.resnets = nn.ModuleList(resnets)
        if with_attention:
            self.attentions = nn.ModuleList([
                Transformer1DModel(heads, out_channels // heads, out_channels, cross_attention_dim, groups)
                for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample1D(out_channels)]) if add_upsample else None

    def forward(self, h, skips, temb, context=None, bias=None, upsample_size=None):
        for i, res in enumerate(self.resnets):
            h = res(torch.cat([h, skips[-1]], dim=1), temb)
            skips = skips[:-1]
            if self.has_cross_attention:
                h = self.attentions[i](h, context, bias)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h, upsample_size)
        return h
