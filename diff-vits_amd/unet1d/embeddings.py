"""Embedding modules of the 1-D conditional UNet (host-side mirror of the reference
interface `unet1d/embeddings.py`: same class names, constructor arguments and parameter
names, so reference checkpoints load unchanged).

`TextTimeEmbedding` is also imported directly by the reference's prior
(reference model3.py:40, 744), which is why it is a public symbol here.
These torch forwards are the `backend="torch"` path (training / CPU use); on a GPU the
UNet runs them inside the HIP engine instead (csrc/engine.hip: the hoisted conditioning of `dv_unet_set_cond`,
csrc/kernels_misc.hip: k_timestep_sincos / k_small_linear(_t) / k_mean_token / k_pool_attn / k_ln_rows).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1,
                           scale=1, max_period=10000):
    """Sinusoidal embedding of (possibly fractional) timesteps [N] -> [N, embedding_dim]
    (reference unet1d/embeddings.py:24-64)."""
    assert timesteps.dim() == 1, "Timesteps should be a 1d-array"
    half = embedding_dim // 2
    freqs = torch.exp(
        -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
        / (half - downscale_freq_shift)
    )
    args = scale * timesteps[:, None].float() * freqs[None, :]
    parts = [torch.cos(args), torch.sin(args)] if flip_sin_to_cos else [torch.sin(args), torch.cos(args)]
    emb = torch.cat(parts, dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    """Parameter-free sinusoidal projection (reference unet1d/embeddings.py:204-218)."""

    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos,
                                      self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    """Linear -> SiLU -> Linear (reference unet1d/embeddings.py:157-201).  Only the
    configuration the sampling path uses (SiLU, no condition projection, no post-act)."""

    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None,
                 cond_proj_dim=None):
        super().__init__()
        if act_fn not in ("silu", "swish") or post_act_fn is not None or cond_proj_dim is not None:
            raise ValueError("TimestepEmbedding: only act_fn='silu' without post-act/cond-proj is supported")
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


class AttentionPooling(nn.Module):
    """Single-query attention pooling over [mean-token, x] (reference
    unet1d/embeddings.py:499-546).  The padding of `x` is not masked, as in the reference."""

    def __init__(self, num_heads, embed_dim, dtype=None):
        super().__init__()
        self.dtype = dtype
        self.positional_embedding = nn.Parameter(torch.randn(1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim, dtype=dtype)
        self.q_proj = nn.Linear(embed_dim, embed_dim, dtype=dtype)
        self.v_proj = nn.Linear(embed_dim, embed_dim, dtype=dtype)
        self.num_heads = num_heads
        self.dim_per_head = embed_dim // num_heads

    def forward(self, x):
        bs, _, width = x.shape
        nh, dph = self.num_heads, self.dim_per_head
        token = x.mean(dim=1, keepdim=True) + self.positional_embedding.to(x.dtype)
        seq = torch.cat([token, x], dim=1)
        q = self.q_proj(token).view(bs, 1, nh, dph).transpose(1, 2)          # [bs, nh, 1, dph]
        k = self.k_proj(seq).view(bs, -1, nh, dph).transpose(1, 2)           # [bs, nh, S, dph]
        v = self.v_proj(seq).view(bs, -1, nh, dph).transpose(1, 2)
        scale = 1.0 / math.sqrt(math.sqrt(dph))
        w = torch.matmul(q * scale, (k * scale).transpose(-1, -2))           # [bs, nh, 1, S]
        w = torch.softmax(w.float(), dim=-1).type(w.dtype)
        return torch.matmul(w, v).reshape(bs, width)


class TextTimeEmbedding(nn.Module):
    """LayerNorm -> AttentionPooling -> Linear -> LayerNorm (reference
    unet1d/embeddings.py:421-434); the `addition_embed_type='text'` branch of the UNet."""

    def __init__(self, encoder_dim, time_embed_dim, num_heads=64):
        super().__init__()
        self.norm1 = nn.LayerNorm(encoder_dim)
        self.pool = AttentionPooling(num_heads, encoder_dim)
        self.proj = nn.Linear(encoder_dim, time_embed_dim)
        self.norm2 = nn.LayerNorm(time_embed_dim)

    def forward(self, hidden_states):
        return self.norm2(self.proj(self.pool(self.norm1(hidden_states))))
