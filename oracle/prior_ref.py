"""CPU restatement of the inference side of the reference's VITS prior, from the text encoder's outputs onward
(VITS.infer, model3.py:817-860): reference encoder (TextTimeEmbedding with one head), DurationPredictor_unet
(model3.py:275-321: 1x1 convs around the UNet at the duration-predictor configuration, integer timestep 1, float
[B,1,L] mask), duration -> alignment path (commons.generate_path, commons.py:128-143), prior sample and the 6-layer
`o_proj` PromptEncoder with speaker conditioning (model3.py:753, 408-412).

TEST INFRASTRUCTURE ONLY.  The text encoder itself (`enc_p`: attentions.Encoder with relative-position attention) is
not restated: tests feed its captured outputs.  Pinned by tools/make_golden_prompt.py against the stub-imported
reference's `vits.infer`: tests/golden/prior_infer.npz.
"""
import torch
import torch.nn.functional as F

from . import prompt_ref, unet_ref

DURPRED_BLOCKS = lambda h: (h // 4, h // 4, h // 2, h // 2)      # noqa: E731  (model3.py:296)


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def ref_enc(sd, y):
    """model3.py:744, 831: TextTimeEmbedding(100, gin, num_heads=1) on y^T -> [B, gin]."""
    e = {"add_embedding." + k: v for k, v in _sub(sd, "ref_enc.").items()}
    return unet_ref.text_time_embedding(e, {"addition_embed_type_num_heads": 1}, y.transpose(1, 2))


def duration_predictor(sd, x, x_lengths, prompt, prompt_lengths, n_heads=8):
    """DurationPredictor_unet.forward, model3.py:304-321."""
    d = _sub(sd, "dp.")
    hidden = d["pre.weight"].shape[0]
    prompt = F.conv1d(prompt, d["prompt_proj.weight"], d["prompt_proj.bias"])
    x_mask = prompt_ref.sequence_mask(x_lengths, x.shape[2]).unsqueeze(1).to(x.dtype)
    prompt_mask = prompt_ref.sequence_mask(prompt_lengths, prompt.shape[2]).unsqueeze(1).to(x.dtype)
    x = F.conv1d(x, d["pre.weight"], d["pre.bias"]) * x_mask
    prompt = prompt * prompt_mask
    cfg = unet_ref.default_config(d["enc.conv_in.weight"].shape[1], 1, DURPRED_BLOCKS(hidden), hidden, n_heads, 8, 2, 64)
    y = unet_ref.unet_forward(_sub(d, "enc."), cfg, x, 1, prompt.transpose(1, 2), prompt_mask)
    return y * x_mask


def generate_path(duration, mask):
    """commons.py:128-143.  duration [b, 1, t_x], mask [b, 1, t_y, t_x]."""
    b, _, t_y, t_x = mask.shape
    cum = torch.cumsum(duration, -1).view(b * t_x)
    path = prompt_ref.sequence_mask(cum, t_y).to(mask.dtype).view(b, t_x, t_y)
    path = path - F.pad(path, (0, 0, 1, 0, 0, 0))[:, :-1]
    return path.unsqueeze(1).transpose(2, 3) * mask


def infer_from_encoder(sd, x, m_p, logs_p, x_mask, x_lengths, y, y_lengths, noise_fn, noise_scale=0.667, length_scale=1):
    """model3.py:831-860 after the `enc_p` call.  noise_fn(shape) supplies the prior noise (the reference draws
    torch.randn_like).  Returns (z, y, y_lengths_out, logw)."""
    g = ref_enc(sd, y).unsqueeze(-1)
    logw = duration_predictor(sd, x, x_lengths, y, y_lengths)
    w = torch.exp(logw) * x_mask * length_scale
    w_ceil = torch.ceil(w)
    y_len = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    y_mask = prompt_ref.sequence_mask(y_len, int(y_len.max())).unsqueeze(1).to(x_mask.dtype)
    attn = generate_path(w_ceil, x_mask.unsqueeze(2) * y_mask.unsqueeze(-1))
    m_p = torch.matmul(attn.squeeze(1), m_p.transpose(1, 2)).transpose(1, 2)
    logs_p = torch.matmul(attn.squeeze(1), logs_p.transpose(1, 2)).transpose(1, 2)
    z_p = m_p + noise_fn(tuple(m_p.shape)) * torch.exp(logs_p) * noise_scale
    o = _sub(sd, "o_proj.")
    gz = F.conv1d(g, o["g_proj.weight"], o["g_proj.bias"])                        # model3.py:410-412
    z = prompt_ref.prompt_encoder(o, z_p + gz, y_len, n_layers=6, num_heads=8)
    return z, y, y_len, logw
