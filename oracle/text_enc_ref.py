"""CPU restatement of the reference's text encoder `enc_p` (model3.py:322-381 TextEncoder over attentions.Encoder,
attentions.py:37-88; MultiHeadAttention with windowed relative-position keys/values :142-300; FFN :322-380).

TEST INFRASTRUCTURE ONLY.  The relative-position terms are restated as explicit band gathers (offset j - i within
+-window, heads share the embeddings) instead of the reference's pad-and-reshape skewing; the arithmetic is the same
products and sums.  Pinned by tools/make_golden_prompt.py: tests/golden/prior_infer.npz (enc_* arrays).
"""
import math

import torch
import torch.nn.functional as F

from . import prompt_ref


def _ln_c(x, gamma, beta, eps=1e-5):
    """attentions.LayerNorm (:12-24): LayerNorm over the channel axis of [B, C, T]."""
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), gamma, beta, eps).transpose(1, -1)


def rel_attention(sd, p, x, attn_mask, n_heads, window):
    """MultiHeadAttention.forward/attention (:179-222) for self-attention with window_size relative embeddings."""
    b, c, t = x.shape
    d = c // n_heads
    q = F.conv1d(x, sd[p + "conv_q.weight"], sd[p + "conv_q.bias"])
    k = F.conv1d(x, sd[p + "conv_k.weight"], sd[p + "conv_k.bias"])
    v = F.conv1d(x, sd[p + "conv_v.weight"], sd[p + "conv_v.bias"])
    q = q.view(b, n_heads, d, t).transpose(2, 3) / math.sqrt(d)
    k = k.view(b, n_heads, d, t).transpose(2, 3)
    v = v.view(b, n_heads, d, t).transpose(2, 3)
    scores = torch.matmul(q, k.transpose(-2, -1))
    i = torch.arange(t).unsqueeze(1)
    j = torch.arange(t).unsqueeze(0)
    off = j - i + window                                             # index into the 2w+1 relative embeddings
    band = (off >= 0) & (off <= 2 * window)
    ek, ev = sd[p + "emb_rel_k"][0], sd[p + "emb_rel_v"][0]         # heads_share: [1, 2w+1, d]
    q_rel = torch.matmul(q, ek.t())                                  # [b, h, t, 2w+1]
    idx = off.clamp(0, 2 * window).expand(b, n_heads, t, t)
    scores = scores + q_rel.gather(-1, idx) * band
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    p_attn = F.softmax(scores, dim=-1)
    out = torch.matmul(p_attn, v)
    r = torch.arange(2 * window + 1).unsqueeze(0)
    jj = i + r - window                                              # key position of relative offset r for query i
    valid = (jj >= 0) & (jj < t)
    rel_w = p_attn.gather(-1, jj.clamp(0, t - 1).expand(b, n_heads, t, 2 * window + 1)) * valid
    out = out + torch.matmul(rel_w, ev)
    out = out.transpose(2, 3).contiguous().view(b, c, t)
    return F.conv1d(out, sd[p + "conv_o.weight"], sd[p + "conv_o.bias"])


def ffn(sd, p, x, x_mask, kernel_size):
    """attentions.FFN.forward (:345-353), 'same' padding, ReLU."""
    pad = ((kernel_size - 1) // 2, kernel_size // 2)
    h = F.conv1d(F.pad(x * x_mask, pad), sd[p + "conv_1.weight"], sd[p + "conv_1.bias"])
    h = torch.relu(h)
    h = F.conv1d(F.pad(h * x_mask, pad), sd[p + "conv_2.weight"], sd[p + "conv_2.bias"])
    return h * x_mask


def encoder(sd, p, x, x_mask, g, n_heads, n_layers, kernel_size, window=4, cond_layer_idx=2):
    """attentions.Encoder.forward (:70-88): speaker embedding added before layer `cond_layer_idx`; post-norm blocks."""
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    x = x * x_mask
    for i in range(n_layers):
        if i == cond_layer_idx and g is not None:
            gl = F.linear(g.transpose(1, 2), sd[p + "spk_emb_linear.weight"], sd[p + "spk_emb_linear.bias"]).transpose(1, 2)
            x = (x + gl) * x_mask
        y = rel_attention(sd, p + "attn_layers.%d." % i, x, attn_mask, n_heads, window)
        x = _ln_c(x + y, sd[p + "norm_layers_1.%d.gamma" % i], sd[p + "norm_layers_1.%d.beta" % i])
        y = ffn(sd, p + "ffn_layers.%d." % i, x, x_mask, kernel_size)
        x = _ln_c(x + y, sd[p + "norm_layers_2.%d.gamma" % i], sd[p + "norm_layers_2.%d.beta" % i])
    return x * x_mask


def text_encoder(sd, ids, lengths, tone, language, g, n_heads=2, n_layers=6, kernel_size=3, prefix="enc_p."):
    """TextEncoder.forward (model3.py:360-381) -> (x, m, logs, x_mask)."""
    hidden = sd[prefix + "emb.weight"].shape[1]
    x = (F.embedding(ids, sd[prefix + "emb.weight"]) + F.embedding(tone, sd[prefix + "tone_emb.weight"])
         + F.embedding(language, sd[prefix + "language_emb.weight"])) * math.sqrt(hidden)
    x = x.transpose(1, -1)
    x_mask = prompt_ref.sequence_mask(lengths, x.shape[2]).unsqueeze(1).to(x.dtype)
    x = encoder(sd, prefix + "encoder.", x * x_mask, x_mask, g, n_heads, n_layers, kernel_size)
    stats = F.conv1d(x, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"]) * x_mask
    m, logs = torch.split(stats, stats.shape[1] // 2, dim=1)
    return x, m, logs, x_mask
