"""GPU parity of each HIP kernel through the C ABI (dv_op_*), against plain fp32 torch-CPU ops
on the same seeded inputs.  Tolerances: bf16x3 contractions 1e-4 relative L2 (measured ~1e-5,
SURVEY.md §7); fp32-MFMA attention and statistics 1e-5."""
import ctypes as C

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _lib():
    from diff_vits_amd import _lib as L
    return L


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _synth(name, shape, scale=1.0):
    from diff_vits_amd import synth
    return synth.normal(99, name, shape) * np.float32(scale)


@pytest.mark.parametrize("B,Cin,T,Cout,k,stride,up_T", [
    (2, 64, 50, 96, 3, 1, 0),        # plain k3, ragged M
    (1, 208, 256, 128, 3, 1, 0),     # conv_in shape: Cin padded 208 -> 224
    (2, 128, 64, 128, 1, 1, 0),      # 1x1
    (2, 96, 37, 96, 3, 2, 0),        # stride-2 downsample, odd T
    (2, 64, 19, 64, 3, 1, 37),       # nearest upsample to explicit size, then k3
    (2, 64, 20, 64, 3, 1, 40),       # x2 upsample
    (8, 128, 1024, 128, 3, 1, 0),    # BASELINE config-2 level-0 shape (128x128 tiles)
    (1, 512, 16, 80, 3, 1, 0),       # N not a multiple of the tile (conv_out-like)
])
@pytest.mark.parametrize("prec", [0, 1])
def test_conv1d(B, Cin, T, Cout, k, stride, up_T, prec):
    L = _lib()
    x = _synth("x", (B, Cin, T))
    w = _synth("w", (Cout, Cin, k), 1.0 / np.sqrt(Cin * k))
    b = _synth("b", (Cout,), 0.1)
    xt = torch.from_numpy(x)
    if up_T:
        xt = F.interpolate(xt, size=up_T, mode="nearest")
    ref = F.conv1d(xt, torch.from_numpy(w), torch.from_numpy(b), stride=stride, padding=(k - 1) // 2)
    y = torch.empty(ref.shape, device="cuda")
    dx, dw, db = _dev(x), _dev(w), _dev(b)      # keep the device tensors alive across the call
    L.check(L.lib().dv_op_conv1d(L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(y), B, Cin, T, Cout, k, stride,
                                 up_T, prec, None), "dv_op_conv1d")
    torch.cuda.synchronize()
    err = rel_l2(y.cpu().numpy(), ref.numpy())
    assert err < (1e-4 if prec == 0 else 2e-2), err


@pytest.mark.parametrize("M,K,N", [(100, 128, 384), (8192, 128, 1024), (1024, 2048, 512), (33, 32, 17)])
def test_linear(M, K, N):
    L = _lib()
    x, w, b = _synth("x", (M, K)), _synth("w", (N, K), 1 / np.sqrt(K)), _synth("b", (N,), 0.1)
    ref = F.linear(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b))
    y = torch.empty((M, N), device="cuda")
    dx, dw, db = _dev(x), _dev(w), _dev(b)
    L.check(L.lib().dv_op_linear(L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(y), M, K, N, 0, None), "dv_op_linear")
    torch.cuda.synchronize()
    assert rel_l2(y.cpu().numpy(), ref.numpy()) < 1e-4


def _split_planes(y):
    """hi = bf16(y), lo = bf16(y - hi) as uint16 bit patterns (round to nearest even, like v_cvt_pk_bf16_f32)."""
    t = torch.from_numpy(np.ascontiguousarray(y))
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi.view(torch.int16).numpy().view(np.uint16), lo.view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize("M,K,N,ldo", [
    # N % 8 != 0 and / or a plane row pitch that is not a multiple of 8 elements (16 bytes): the 16-byte plane stores of the
    # epilogue (store_planes16 / store_planes8: lane-pair exchange, uint4 stores) must NOT be taken - VERDICT r3 #8: the
    # alignment guard of commit 425603b had no test; before it these cases faulted or tore rows
    (100, 64, 17, 17), (100, 64, 17, 19), (70, 128, 100, 100), (70, 128, 100, 104), (70, 128, 100, 108), (200, 64, 132, 132),
    (200, 64, 132, 140), (129, 256, 128, 132), (129, 256, 128, 130), (64, 64, 64, 68),
    # ... and the aligned shapes that DO take them (whole fragments, half-fragment epilogue of the two-k-group tiles)
    (128, 256, 128, 128), (4096, 128, 256, 256), (300, 64, 96, 104), (2048, 384, 384, 384),
])
def test_linear_planes_any_pitch(M, K, N, ldo):
    """Split-plane outputs of the GEMM epilogue at every alignment class of (N, pitch): bit-equal to the planes split on the
    host from the SAME launch's fp32 output, the fp32 output equal to the plain fp32 launch, and not a byte written past a
    row's N columns."""
    L = _lib()
    x, w, b = _synth("x", (M, K)), _synth("w", (N, K), 1 / np.sqrt(K)), _synth("b", (N,), 0.1)
    dx, dw, db = _dev(x), _dev(w), _dev(b)
    y0 = torch.empty((M, N), device="cuda")
    L.check(L.lib().dv_op_linear(L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(y0), M, K, N, 0, None), "dv_op_linear")
    fill = 0x7fc1                                                     # a NaN pattern no result can take
    for with_f32 in (True, False):
        y = torch.full((M, ldo), float("nan"), device="cuda")
        hi = torch.full((M, ldo), fill, dtype=torch.int32, device="cuda").to(torch.int16)
        lo = hi.clone()
        L.check(L.lib().dv_op_linear_planes(L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(y) if with_f32 else None, L.ptr(hi), L.ptr(lo),
                                            M, K, N, ldo, 0, 0, None), "dv_op_linear_planes")
        torch.cuda.synchronize()
        h = hi.cpu().numpy().view(np.uint16); l = lo.cpu().numpy().view(np.uint16)
        ref = y0.cpu().numpy()
        if with_f32:
            yy = y.cpu().numpy()
            assert np.array_equal(yy[:, :N], ref)                     # same tile, same summation order: bit-equal
            assert np.isnan(yy[:, N:]).all()
        rh, rl = _split_planes(ref)
        assert np.array_equal(h[:, :N], rh), "hi plane differs at %s" % (np.argwhere(h[:, :N] != rh)[:4],)
        assert np.array_equal(l[:, :N], rl), "lo plane differs at %s" % (np.argwhere(l[:, :N] != rl)[:4],)
        assert (h[:, N:] == fill).all() and (l[:, N:] == fill).all(), "bytes written past a row's N columns"


@pytest.mark.parametrize("M,K,No,ldo", [(96, 64, 96, 96), (96, 64, 96, 100), (200, 128, 160, 164), (130, 64, 32, 36), (4096, 256, 1024, 1024)])
def test_linear_planes_geglu_any_pitch(M, K, No, ldo):
    """GEGLU epilogue (value * gelu(gate), reference unet1d/attention.py GEGLU) -> split planes at aligned and unaligned
    pitches, against the fp64 product split on the host (within the split-bf16 contraction's error) and, for the unaligned
    pitches, bit-equal to the aligned launch's planes (the store path must not change a value)."""
    L = _lib()
    x, w, b = _synth("x", (M, K)), _synth("w", (2 * No, K), 1 / np.sqrt(K)), _synth("b", (2 * No,), 0.1)
    dx, dw, db = _dev(x), _dev(w), _dev(b)
    z = torch.from_numpy(x).double() @ torch.from_numpy(w).double().t() + torch.from_numpy(b).double()
    ref = (z[:, :No] * F.gelu(z[:, No:])).numpy()
    outs = {}
    for pitch in sorted({No, ldo}):
        hi = torch.full((M, pitch), 0x7fc1, dtype=torch.int32, device="cuda").to(torch.int16)
        lo = hi.clone()
        L.check(L.lib().dv_op_linear_planes(L.ptr(dx), L.ptr(dw), L.ptr(db), None, L.ptr(hi), L.ptr(lo), M, K, 2 * No, pitch, 1, 0, None),
                "dv_op_linear_planes(geglu)")
        torch.cuda.synchronize()
        h = hi.cpu().numpy().view(np.uint16); l = lo.cpu().numpy().view(np.uint16)
        assert (h[:, No:] == 0x7fc1).all() and (l[:, No:] == 0x7fc1).all()
        val = (torch.from_numpy(h[:, :No].copy().view(np.int16)).view(torch.bfloat16).double()
               + torch.from_numpy(l[:, :No].copy().view(np.int16)).view(torch.bfloat16).double()).numpy()
        assert rel_l2(val, ref) < 3e-5
        outs[pitch] = (h[:, :No].copy(), l[:, :No].copy())
    if ldo != No:
        assert np.array_equal(outs[No][0], outs[ldo][0]) and np.array_equal(outs[No][1], outs[ldo][1])


@pytest.mark.parametrize("B,T,C,G", [(2, 40, 32, 8), (8, 1024, 128, 8), (2, 100, 896, 8), (3, 7, 1024, 8), (2, 300, 96, 8)])
def test_group_stats(B, T, C, G):
    L = _lib()
    x = _synth("x", (B, T, C)) + np.float32(0.7)
    mean = torch.empty((B, G), device="cuda")
    rstd = torch.empty((B, G), device="cuda")
    dx = _dev(x)
    L.check(L.lib().dv_op_group_stats(L.ptr(dx), L.ptr(mean), L.ptr(rstd), B, T, C, G, 1e-5, None), "dv_op_group_stats")
    torch.cuda.synchronize()
    xg = x.astype(np.float64).reshape(B, T, G, C // G)
    m = xg.mean(axis=(1, 3))
    v = xg.var(axis=(1, 3))
    assert np.abs(mean.cpu().numpy() - m).max() < 1e-5
    assert rel_l2(rstd.cpu().numpy(), 1 / np.sqrt(v + 1e-5)) < 1e-5


@pytest.mark.parametrize("B,H,Tq,Tk,d,masked", [
    (2, 8, 40, 40, 4, False), (2, 8, 100, 50, 16, True), (1, 8, 1024, 1024, 16, False), (2, 8, 512, 256, 32, True),
    (2, 8, 256, 256, 48, False), (2, 8, 128, 256, 64, True), (2, 8, 37, 60, 8, True), (1, 8, 64, 33, 12, False),
    # odd numbers of 32-key sub-tiles (the trailing one is processed fully masked), with 1..8 waves per workgroup
    (1, 8, 96, 70, 16, True), (2, 8, 33, 129, 32, False), (1, 8, 300, 97, 48, True), (2, 8, 512, 80, 16, True),
    (2, 8, 512, 161, 64, False),
])
def test_attention(B, H, Tq, Tk, d, masked):
    L = _lib()
    q, k, v = _synth("q", (B, Tq, H * d)), _synth("k", (B, Tk, H * d)), _synth("v", (B, Tk, H * d))
    bias = None
    if masked:
        bias = np.zeros((B, Tk), dtype=np.float32)
        for b in range(B):
            bias[b, max(1, Tk - 7 * (b + 1)):] = -10000.0
    tq = torch.from_numpy(q).view(B, Tq, H, d).transpose(1, 2)
    tk = torch.from_numpy(k).view(B, Tk, H, d).transpose(1, 2)
    tv = torch.from_numpy(v).view(B, Tk, H, d).transpose(1, 2)
    am = None if bias is None else torch.from_numpy(bias)[:, None, None, :].expand(B, H, 1, Tk)
    ref = F.scaled_dot_product_attention(tq.double(), tk.double(), tv.double(),
                                         attn_mask=None if am is None else am.double()).transpose(1, 2).reshape(B, Tq, H * d)
    o = torch.empty((B, Tq, H * d), device="cuda")
    dq, dk, dv = _dev(q), _dev(k), _dev(v)
    dbias = None if bias is None else _dev(bias)
    L.check(L.lib().dv_op_attention(L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(dbias), L.ptr(o), B, H, Tq, Tk, d, None),
            "dv_op_attention")
    torch.cuda.synchronize()
    # Round 4: the probabilities enter P V as ONE fp16 plane (11 significant bits, round to nearest even; V and both Q K^T
    # operands stay split).  With zero-mean random V the rounding errors of P do not average out against the result (both are
    # random-walk sums over the keys): a single attention sits at the rounding's own rms, 2^-12 / sqrt(3) = 1.4e-4 (measured
    # 1.6-1.7e-4 on every shape here; 3e-6 with split-bf16 P).  Inside the denoiser the effect is 1.3e-5 -> 1.8e-5 of the
    # reference output (profiles/r04_err_vs_goldens_p_fp16.txt; budget 1e-3).  DV_ATTN_PF16=0 builds restore the old form.
    err = rel_l2(o.cpu().numpy(), ref.numpy())
    assert err < 3e-4, err
