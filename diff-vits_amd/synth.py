"""Deterministic synthetic weights and inputs for the diffusion-sampling path.

No checkpoint ships with the reference, so parity and the bench
run on seeded synthetic tensors.  Everything here is integer hashing followed by
exact float arithmetic, so the same (seed, name, shape) gives bit-identical
values in the build container and on the GPU box, with no dependence on libm or
on a random-number library.
"""
import zlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    """splitmix64 finaliser on a uint64 array."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        x = x ^ (x >> np.uint64(31))
    return x


def _stream(seed, name, n, lanes=1):
    """`lanes` independent uint64 hash streams of length n for (seed, name)."""
    tag = np.uint64(zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF)
    base = _mix(np.uint64(seed) ^ (tag << np.uint64(32)))
    idx = np.arange(n * lanes, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _mix(idx * np.uint64(0xD6E8FEB86659FD93) + base)
    return h.reshape(n, lanes)


def uniform(seed, name, shape, bound=1.0):
    """Uniform in [-bound, bound): 24-bit dyadic fractions scaled by `bound` (float32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = _stream(seed, name, n)[:, 0]
    u = ((h >> np.uint64(40)).astype(np.float64) / float(1 << 24)) * 2.0 - 1.0
    return (u * float(bound)).astype(np.float32).reshape(shape)


def normal(seed, name, shape, std=1.0):
    """Approximately N(0, std^2): Irwin-Hall sum of 12 uniforms (exact dyadic arithmetic)."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = _stream(seed, name, n, lanes=12)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    z = u.sum(axis=1) - 6.0
    return (z * float(std)).astype(np.float32).reshape(shape)


def make_state_dict(shapes, seed=1234):
    """Seeded weights for a {name: shape} table, in PyTorch-default-like ranges.

    Conv/Linear weights: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); biases the same
    bound; norm weights 1 + U(-0.1, 0.1), norm biases U(-0.1, 0.1); the
    attention-pooling positional embedding N(0, 1/embed_dim).
    """
    out = {}
    weight_fan = {}
    for name, shape in shapes.items():
        if name.endswith(".weight") and len(shape) >= 2:
            weight_fan[name[: -len(".weight")]] = int(np.prod(shape[1:]))
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        stem, _, leaf = name.rpartition(".")
        is_norm = ("norm" in stem.split(".")[-1]) and len(shape) == 1
        if leaf == "positional_embedding":
            v = normal(seed, name, shape, std=1.0 / np.sqrt(shape[-1]))
        elif is_norm and leaf == "weight":
            v = 1.0 + uniform(seed, name, shape, 0.1)
        elif is_norm and leaf == "bias":
            v = uniform(seed, name, shape, 0.1)
        elif leaf == "weight":
            v = uniform(seed, name, shape, 1.0 / np.sqrt(int(np.prod(shape[1:]))))
        elif leaf == "bias":
            fan = weight_fan.get(stem, shape[0])
            v = uniform(seed, name, shape, 1.0 / np.sqrt(fan))
        else:
            v = uniform(seed, name, shape, 0.1)
        out[name] = v.astype(np.float32)
    return out


def make_inputs(B, C, T, L, cond_channels=128, enc_dim=128, seed=1234, ragged_mask=False):
    """Synthetic sampler inputs: x_T, cond, enc ~ N(0,1); mask bool [B,L]."""
    x = normal(seed, "x_T", (B, C, T))
    cond = normal(seed, "cond", (B, cond_channels, T))
    enc = normal(seed, "enc", (B, L, enc_dim))
    mask = np.ones((B, L), dtype=bool)
    if ragged_mask:
        for b in range(B):
            mask[b, max(1, L - 16 * b):] = False
    return x, cond, enc, mask


def make_betas(n=1000, beta_start=1e-4, beta_end=2e-2):
    """Linear beta schedule as the reference builds it: float64 linspace cast to float32
    (reference model3.py:935-942, 990)."""
    return np.linspace(beta_start, beta_end, n, dtype=np.float64).astype(np.float32)
