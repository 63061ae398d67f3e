"""ORACLE (test infrastructure, not product code): CPU restatement of the reference
denoiser `unet1d.unet_1d_condition.UNet1DConditionModel.forward`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  It is a functional PyTorch-CPU restatement over a flat
{parameter-name: tensor} dict that uses the reference's parameter names, so the
same seeded state dict can be pushed into the imported reference (here, in the
build container) and into this file.  Pinned against the imported reference by
tools/make_golden.py -> tests/golden/*.npz (the reference has no tests of its
own for this path: SURVEY.md §4, §8c).

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  The arithmetic primitives (conv1d, group_norm, layer_norm,
SDPA-equivalent softmax attention, gelu-erf, nearest interpolate) are torch CPU
ops, exactly the ops the reference itself calls.
"""
import math

import torch
import torch.nn.functional as F


def default_config(in_channels=208, out_channels=80, block_out_channels=(128, 256, 384, 512),
                   cross_attention_dim=128, num_heads=8, norm_num_groups=8, layers_per_block=2,
                   addition_embed_type_num_heads=64, norm_eps=1e-5):
    """Diffusion-encoder ctor kwargs (reference model3.py:887-896) + ctor defaults
    (unet1d/unet_1d_condition.py:151-203)."""
    return dict(in_channels=in_channels, out_channels=out_channels,
                block_out_channels=tuple(block_out_channels), cross_attention_dim=cross_attention_dim,
                num_heads=num_heads, norm_num_groups=norm_num_groups, layers_per_block=layers_per_block,
                addition_embed_type_num_heads=addition_embed_type_num_heads, norm_eps=norm_eps)


# ----------------------------------------------------------------------------- embeddings
def timestep_embedding(timesteps, dim):
    """unet1d/embeddings.py:24-64 with flip_sin_to_cos=True, downscale_freq_shift=0
    (unet_1d_condition.py:275): [cos | sin] of t * exp(-ln(1e4) k / half)."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def time_mlp(sd, t_emb):
    """TimestepEmbedding.forward, unet1d/embeddings.py:186-201 (act = SiLU)."""
    h = F.linear(t_emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    h = F.silu(h)
    return F.linear(h, sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])


def text_time_embedding(sd, cfg, enc):
    """TextTimeEmbedding + AttentionPooling, unet1d/embeddings.py:421-434, 499-546.
    No padding mask is applied."""
    p = "add_embedding."
    x = F.layer_norm(enc, (enc.shape[-1],), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    bs, length, width = x.shape
    nh = cfg["addition_embed_type_num_heads"]
    dph = width // nh
    cls = x.mean(dim=1, keepdim=True) + sd[p + "pool.positional_embedding"].to(x.dtype)
    xx = torch.cat([cls, x], dim=1)

    def heads(v):  # (bs, n, width) -> (bs*nh, dph, n)
        return v.view(bs, -1, nh, dph).transpose(1, 2).reshape(bs * nh, -1, dph).transpose(1, 2)

    q = heads(F.linear(cls, sd[p + "pool.q_proj.weight"], sd[p + "pool.q_proj.bias"]))
    k = heads(F.linear(xx, sd[p + "pool.k_proj.weight"], sd[p + "pool.k_proj.bias"]))
    v = heads(F.linear(xx, sd[p + "pool.v_proj.weight"], sd[p + "pool.v_proj.bias"]))
    scale = 1 / math.sqrt(math.sqrt(dph))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v)
    a = a.reshape(bs, -1, 1).transpose(1, 2)[:, 0, :]
    a = F.linear(a, sd[p + "proj.weight"], sd[p + "proj.bias"])
    return F.layer_norm(a, (a.shape[-1],), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)


# ----------------------------------------------------------------------------- blocks
def resnet_block(sd, p, cfg, x, emb):
    """ResnetBlock2D.forward with time_embedding_norm='scale_shift',
    unet1d/resnet.py:591-641."""
    g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    h = F.group_norm(x, g, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    h = F.silu(h)
    h = F.conv1d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    t = F.linear(F.silu(emb), sd[p + "time_emb_proj.weight"], sd[p + "time_emb_proj.bias"])[:, :, None]
    h = F.group_norm(h, g, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    scale, shift = torch.chunk(t, 2, dim=1)
    h = h * (1 + scale) + shift
    h = F.silu(h)
    h = F.conv1d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if (p + "conv_shortcut.weight") in sd:
        x = F.conv1d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def attention(sd, p, heads, x, ctx=None, bias=None):
    """Attention + AttnProcessor2_0.__call__, unet1d/attention_processor.py:971-1052.
    `bias` is the additive [B,1,L] mask bias; it is broadcast over heads and queries
    (prepare_attention_mask :309-336 -> view [B,H,1,L])."""
    B = x.shape[0]
    src = x if ctx is None else ctx
    q = F.linear(x, sd[p + "to_q.weight"])
    k = F.linear(src, sd[p + "to_k.weight"])
    v = F.linear(src, sd[p + "to_v.weight"])
    d = q.shape[-1] // heads
    q = q.view(B, -1, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    mask = None if bias is None else bias[:, None, :, :].expand(B, heads, 1, bias.shape[-1])
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(B, -1, heads * d)
    return F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def transformer_block(sd, p, heads, x, enc, bias):
    """BasicTransformerBlock.forward, unet1d/attention.py:130-203 (LayerNorm eps 1e-5,
    GEGLU feed-forward :280-301 with exact-erf GELU)."""
    C = x.shape[-1]
    n = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    x = attention(sd, p + "attn1.", heads, n) + x
    n = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    x = attention(sd, p + "attn2.", heads, n, enc, bias) + x
    n = F.layer_norm(x, (C,), sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5)
    hg = F.linear(n, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"])
    a, gate = hg.chunk(2, dim=-1)
    ff = F.linear(a * F.gelu(gate), sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])
    return ff + x


def transformer_1d(sd, p, cfg, x, enc, bias):
    """Transformer2DModel.forward (continuous input, conv projections),
    unet1d/transformer_1d.py:191-326: GN(eps 1e-6) -> 1x1 conv -> (B,T,C) -> block ->
    (B,C,T) -> 1x1 conv -> + residual."""
    res = x
    h = F.group_norm(x, cfg["norm_num_groups"], sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    h = F.conv1d(h, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    h = h.permute(0, 2, 1)
    h = transformer_block(sd, p + "transformer_blocks.0.", cfg["num_heads"], h, enc, bias)
    h = h.permute(0, 2, 1).contiguous()
    h = F.conv1d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return h + res


def downsample(sd, p, x):
    """Downsample2D.forward, unet1d/resnet.py:214-223: Conv1d(k3, s2, p1)."""
    return F.conv1d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=2, padding=1)


def upsample(sd, p, x, size=None):
    """Upsample2D.forward, unet1d/resnet.py:138-173: nearest x2, or to `size` when the
    caller forwards an explicit size; then Conv1d(k3, p1)."""
    if size is None:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    else:
        x = F.interpolate(x, size=size, mode="nearest")
    return F.conv1d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1)


# ----------------------------------------------------------------------------- forward
def unet_forward(sd, cfg, sample, timestep, enc, enc_mask=None, probes=None):
    """UNet1DConditionModel.forward, unet1d/unet_1d_condition.py:743-1037, for the block
    layout down=(CrossAttn x3, Down), mid=CrossAttn, up=(Up, CrossAttn x3)
    (SURVEY.md Appendix A).

    sample [B,Cin,T]; timestep float/int tensor [B] or scalar; enc [B,L,D];
    enc_mask bool/0-1 [B,L] or additive bias [B,1,L].  Returns [B,Cout,T].
    `probes`, when a dict, receives named intermediate tensors.
    """
    chans = cfg["block_out_channels"]
    nlev = len(chans)
    lpb = cfg["layers_per_block"]
    B = sample.shape[0]

    # :789-797  (quirk 1: tests both C_in and T)
    up_factor = 2 ** (nlev - 1)
    forward_upsample_size = any(s % up_factor != 0 for s in sample.shape[-2:])

    # :816-818
    bias = None
    if enc_mask is not None:
        if enc_mask.dim() == 2:
            bias = ((1 - enc_mask.to(sample.dtype)) * -10000.0).unsqueeze(1)
        else:
            bias = (1 - enc_mask.to(sample.dtype)) * -10000.0

    # :825-848
    if not torch.is_tensor(timestep):
        timestep = torch.tensor([timestep], dtype=torch.float64 if isinstance(timestep, float) else torch.int64)
    elif timestep.dim() == 0:
        timestep = timestep[None]
    timestep = timestep.expand(B)
    t_emb = timestep_embedding(timestep, chans[0]).to(sample.dtype)
    emb = time_mlp(sd, t_emb)
    # :869-870, :918
    emb = emb + text_time_embedding(sd, cfg, enc)
    if probes is not None:
        probes["emb"] = emb

    # :943
    h = F.conv1d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    if probes is not None:
        probes["conv_in"] = h
    skips = [h]

    # :950-973
    for i in range(nlev):
        bp = "down_blocks.%d." % i
        has_attn = i < nlev - 1
        for j in range(lpb):
            h = resnet_block(sd, bp + "resnets.%d." % j, cfg, h, emb)
            if has_attn:
                h = transformer_1d(sd, bp + "attentions.%d." % j, cfg, h, enc, bias)
            skips.append(h)
        if i < nlev - 1:
            h = downsample(sd, bp + "downsamplers.0.", h)
            skips.append(h)
        if probes is not None:
            probes["down%d" % i] = h

    # :987-995
    h = resnet_block(sd, "mid_block.resnets.0.", cfg, h, emb)
    h = transformer_1d(sd, "mid_block.attentions.0.", cfg, h, enc, bias)
    h = resnet_block(sd, "mid_block.resnets.1.", cfg, h, emb)
    if probes is not None:
        probes["mid"] = h

    # :1001-1026
    for i in range(nlev):
        bp = "up_blocks.%d." % i
        has_attn = i > 0
        is_final = i == nlev - 1
        n_res = lpb + 1
        res = skips[-n_res:]
        skips = skips[:-n_res]
        size = None
        if not is_final and forward_upsample_size:
            size = skips[-1].shape[2:]
        for j in range(n_res):
            h = torch.cat([h, res[-1]], dim=1)
            res = res[:-1]
            h = resnet_block(sd, bp + "resnets.%d." % j, cfg, h, emb)
            if has_attn:
                h = transformer_1d(sd, bp + "attentions.%d." % j, cfg, h, enc, bias)
        if not is_final:
            h = upsample(sd, bp + "upsamplers.0.", h, size)
        if probes is not None:
            probes["up%d" % i] = h

    # :1029-1032
    h = F.group_norm(h, cfg["norm_num_groups"], sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], cfg["norm_eps"])
    h = F.silu(h)
    return F.conv1d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def diffusion_model_fn(sd, cfg, cond, enc, enc_mask):
    """The x0-prediction callable the sampler drives: channel-concat of the noisy mel with
    the content condition, then the UNet (reference model3.py:908-914, without the prompt
    encoder, which is out of this path's scope: SURVEY.md §8f row 1)."""
    def fn(x, t_input):
        return unet_forward(sd, cfg, torch.cat([x, cond], dim=1), t_input, enc, enc_mask)
    return fn
