"""CPU restatement of the reference's PromptEncoder and of the Diffusion_Encoder glue around the denoiser.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
by the product path.  Functional PyTorch-CPU code over a flat {reference-parameter-name: tensor} dict; every
function cites the reference lines it follows.  Pinned by tools/make_golden.py against the stub-imported
reference (`model3.Diffusion_Encoder`) on seeded synthetic weights: tests/golden/prompt_*.npz.

Reference call tree (diff-vits):
  model3.py:867-914   Diffusion_Encoder.__init__/forward  (prompt encoder -> mask -> cat -> UNet)
  model3.py:382-433   PromptEncoder  (ConvLayer pre, n x TransformerEncoderLayer(8), ConvLayer out_proj, LayerNorm)
  model.py:72-81      TransformerEncoderLayer -> operations.OPERATIONS_ENCODER[8] = EncSALayer(c, 8 heads, k=9 'SAME')
  model.py:137-171    ConvTBC / ConvLayer (masked_fill -> LayerNorm -> conv_tbc)
  operations.py:784-821   EncSALayer.forward
  operations.py:304-416   MultiheadAttention (self_attention=True, bias=False -> F.multi_head_attention_forward)
  operations.py:644-693   TransformerFFNLayer (k = 9 as nine shifted Linears, scaled by 9^-1/2)
  commons.py:121-125  sequence_mask
"""
import torch
import torch.nn.functional as F

from . import unet_ref


def sequence_mask(lengths, max_length):
    """commons.py:121-125."""
    x = torch.arange(max_length, dtype=lengths.dtype)
    return x.unsqueeze(0) < lengths.unsqueeze(1)


def conv_layer(sd, p, x, pad_mask=None):
    """model.py:153-171 ConvLayer with kernel_size 1: masked_fill (only when a mask is passed) -> LayerNorm ->
    conv_tbc.  x: [T, B, C_in]; weight [1, C_in, C_out] (model.py:145-146)."""
    if pad_mask is not None:
        x = x.masked_fill(pad_mask.t().unsqueeze(-1), 0)
    x = F.layer_norm(x, (x.shape[-1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
    w = sd[p + "conv.weight"]
    assert w.shape[0] == 1, "only kernel_size 1 is used on this path (model3.py:402-403)"
    return torch.matmul(x, w[0]) + sd[p + "conv.bias"]            # conv_tbc, k = 1, padding 0


def self_attention(sd, p, x, pad_mask, num_heads):
    """operations.py:405-416 -> F.multi_head_attention_forward with the fused in_proj_weight [3C, C], no biases,
    key_padding_mask = -inf on padded keys, q scaled by head_dim^-1/2.  x: [T, B, C]."""
    T, B, C = x.shape
    d = C // num_heads
    qkv = F.linear(x, sd[p + "in_proj_weight"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = q * (d ** -0.5)

    def heads(t):     # [T, B, C] -> [B, H, T, d]
        return t.reshape(T, B, num_heads, d).permute(1, 2, 0, 3)

    q, k, v = heads(q), heads(k), heads(v)
    s = torch.matmul(q, k.transpose(-1, -2))                       # [B, H, T, T]
    s = s.masked_fill(pad_mask[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v).permute(2, 0, 1, 3).reshape(T, B, C)
    return F.linear(o, sd[p + "out_proj.weight"])


def ffn(sd, p, x, kernel_size=9):
    """operations.py:664-693: 'SAME' padding, kernel_size shifted Linears (bias only on the first), sum scaled by
    kernel_size^-1/2, ReLU, ffn_2.  QUIRK kept (operations.py:678): tap 0 multiplies the UNPADDED x, i.e. offset 0
    instead of -(k-1)/2; taps i >= 1 see offsets i - (k-1)/2."""
    T = x.shape[0]
    first = (kernel_size - 1) // 2
    padded = F.pad(x, (0, 0, 0, 0, first, kernel_size - 1 - first))
    res = 0
    for i in range(kernel_size):
        shifted = padded[i:T + i] if i else x
        res = res + F.linear(shifted, sd[p + "ffn_1.%d.weight" % i], sd[p + "ffn_1.0.bias"] if i == 0 else None)
    h = F.relu(res * kernel_size ** -0.5)
    return F.linear(h, sd[p + "ffn_2.weight"], sd[p + "ffn_2.bias"])


def enc_sa_layer(sd, p, x, pad_mask, num_heads=8, kernel_size=9):
    """operations.py:798-821 (dropout is the identity at inference)."""
    keep = (1 - pad_mask.float()).transpose(0, 1)[..., None]      # [T, B, 1]
    res = x
    x = F.layer_norm(x, (x.shape[-1],), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], 1e-5)
    x = res + self_attention(sd, p + "self_attn.", x, pad_mask, num_heads)
    x = x * keep
    res = x
    x = F.layer_norm(x, (x.shape[-1],), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], 1e-5)
    x = res + ffn(sd, p + "ffn.", x, kernel_size)
    return x * keep


def prompt_encoder(sd, prompt, lengths, n_layers=4, num_heads=8, prefix="", probes=None):
    """model3.py:408-433.  prompt [B, C_in, L], lengths [B] -> [B, C_out, L]."""
    x = prompt.permute(2, 0, 1)                                    # b c t -> t b c
    pad = ~sequence_mask(lengths, x.shape[0])                      # [B, L], True = padding
    keep = (1 - pad.float()).transpose(0, 1)[..., None]
    x = conv_layer(sd, prefix + "pre.", x, pad) * keep
    if probes is not None:
        probes["pre"] = x.permute(1, 0, 2).clone()
    for i in range(n_layers):
        x = enc_sa_layer(sd, prefix + "layers.%d.op." % i, x, pad, num_heads)
        if probes is not None:
            probes["layer%d" % i] = x.permute(1, 0, 2).clone()
    x = conv_layer(sd, prefix + "out_proj.", x) * keep             # no mask passed here (model3.py:427)
    if prefix + "layer_norm.weight" in sd:                         # last_ln
        x = F.layer_norm(x, (x.shape[-1],), sd[prefix + "layer_norm.weight"], sd[prefix + "layer_norm.bias"], 1e-5)
        x = x * keep
    return x.permute(1, 2, 0)                                      # t b c -> b c t


def diffusion_encoder_forward(sd, unet_cfg, x, cond, prompt, prompt_lengths, t, n_layers=4, num_heads=8):
    """model3.py:902-914: prompt encoder (state-dict prefix 'prompt_encoder.') -> re-mask -> cat([x, cond]) ->
    UNet (prefix 'unet.') with encoder_hidden_states = prompt^T and the bool sequence mask."""
    pe = {k[len("prompt_encoder."):]: v for k, v in sd.items() if k.startswith("prompt_encoder.")}
    un = {k[len("unet."):]: v for k, v in sd.items() if k.startswith("unet.")}
    mask = sequence_mask(prompt_lengths, prompt.shape[2])
    enc = prompt_encoder(pe, prompt, prompt_lengths, n_layers, num_heads) * mask.unsqueeze(1).to(x.dtype)
    return unet_ref.unet_forward(un, unet_cfg, torch.cat([x, cond], dim=1), t, enc.transpose(1, 2), mask)
