import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import diff_vits_amd  # noqa: E402,F401  (registers the package alias)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="session")
def gold():
    def load(name):
        return np.load(os.path.join(GOLD, name))
    return load


# UNet cases of tools/make_golden.py (kept in sync by tests/test_oracle_golden.py)
UNET_CASES = {
    "tiny": (dict(in_channels=24, out_channels=8, block_out_channels=(32, 64, 96, 128), norm_num_groups=8,
                  cross_attention_dim=32, attention_head_dim=8, addition_embed_type="text",
                  resnet_time_scale_shift="scale_shift", addition_embed_type_num_heads=8), 2, 40, 12, True, "frac"),
    "cfg1": (dict(in_channels=208, out_channels=80, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
                  cross_attention_dim=128, attention_head_dim=8, addition_embed_type="text",
                  resnet_time_scale_shift="scale_shift"), 1, 256, 128, False, "frac"),
    "oddT": (dict(in_channels=208, out_channels=80, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
                  cross_attention_dim=128, attention_head_dim=8, addition_embed_type="text",
                  resnet_time_scale_shift="scale_shift"), 2, 100, 50, True, "frac"),
    "c100": (dict(in_channels=228, out_channels=100, block_out_channels=(128, 256, 384, 512), norm_num_groups=8,
                  cross_attention_dim=128, attention_head_dim=8, addition_embed_type="text",
                  resnet_time_scale_shift="scale_shift"), 2, 64, 40, False, "frac"),
    "durpred": (dict(in_channels=256, out_channels=1, block_out_channels=(64, 64, 128, 128), norm_num_groups=8,
                     cross_attention_dim=256, attention_head_dim=8, addition_embed_type="text",
                     resnet_time_scale_shift="scale_shift"), 2, 37, 60, True, "int1"),
}


def unet_case(name):
    """(ctor kwargs, state dict (numpy), sample, timestep, enc, mask) for a golden case."""
    import torch
    from diff_vits_amd import synth
    from diff_vits_amd.unet1d.unet_1d_condition import UNet1DConditionModel

    kw, B, T, L, ragged, tspec = UNET_CASES[name]
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in UNet1DConditionModel(**kw).state_dict().items()}
    sd = synth.make_state_dict(shapes, seed=1234)
    sample = synth.normal(1234, "sample", (B, kw["in_channels"], T))
    enc = synth.normal(1234, "enc", (B, L, kw["cross_attention_dim"]))
    mask = np.ones((B, L), dtype=bool)
    if ragged:
        for b in range(B):
            mask[b, max(1, L - 5 * (b + 1)):] = False
    t = np.array([949.05 - 37.5 * b for b in range(B)], dtype=np.float32) if tspec == "frac" else 1
    return kw, sd, sample, t, enc, mask


def oracle_cfg(kw):
    from oracle import unet_ref
    return unet_ref.default_config(kw["in_channels"], kw["out_channels"], kw["block_out_channels"],
                                   kw["cross_attention_dim"], kw["attention_head_dim"], kw["norm_num_groups"], 2,
                                   kw.get("addition_embed_type_num_heads", 64))
