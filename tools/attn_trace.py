#!/usr/bin/env python3
"""Per-workgroup phase timeline of the attention kernel (GPU box, development build `make -C diff-vits_amd/csrc trace`).
Thread 0 of each workgroup stamps s_memtime at: entry | Q split | ring filled | first sub-tiles landed | converted |
first iteration done | key loop done | end.  Usage: attn_trace.py [BxHxTqxTkxd ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import diff_vits_amd  # noqa
from diff_vits_amd import _lib as L

L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libdvits_hip_trace.so")
lib = L.lib()
lib.dv_debug_attn_trace.restype = C.c_int
lib.dv_debug_attn_trace.argtypes = [C.c_void_p, C.c_int]

shapes = [(8, 8, 1024, 1024, 16), (8, 8, 1024, 256, 16), (8, 8, 512, 512, 32), (8, 8, 512, 256, 32), (8, 8, 256, 256, 48),
          (8, 8, 128, 128, 64), (8, 8, 128, 256, 64)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
NWG = 4096
buf = np.zeros((NWG, 16), dtype=np.uint64)
names = ["geometry", "fill ring", "Q + first tiles land", "convert 2 tiles", "first iteration", "rest of key loop",
         "normalise + store", "whole workgroup"]
for B, H, Tq, Tk, d in shapes:
    q = torch.randn(B, Tq, H * d, device="cuda")
    k = torch.randn(B, Tk, H * d, device="cuda")
    v = torch.randn(B, Tk, H * d, device="cuda")
    o = torch.empty(B, Tq, H * d, device="cuda")
    for _ in range(3):
        L.check(lib.dv_op_attention(L.ptr(q), L.ptr(k), L.ptr(v), None, L.ptr(o), B, H, Tq, Tk, d, None))
    torch.cuda.synchronize()
    assert lib.dv_debug_attn_trace_clear() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.dv_op_attention(L.ptr(q), L.ptr(k), L.ptr(v), None, L.ptr(o), B, H, Tq, Tk, d, None))
    e1.record()
    torch.cuda.synchronize()
    assert lib.dv_debug_attn_trace(buf.ctypes.data_as(C.c_void_p), NWG) == 0
    t = buf.astype(np.int64)
    t = t[t[:, 7] > 0]
    if Tk <= 64:
        t[:, 5] = t[:, 4]
    ph = np.stack([t[:, i + 1] - t[:, i] for i in range(7)] + [t[:, 7] - t[:, 0]], 1)
    print("B=%d H=%d Tq=%d Tk=%d d=%d  workgroups=%d  op %.1f us  (first start -> last end %d ticks)"
          % (B, H, Tq, Tk, d, len(t), e0.elapsed_time(e1) * 1e3, t[:, 7].max() - t[:, 0].min()))
    for i, nm in enumerate(names):
        print("   %-20s median %7d   p10 %7d   p90 %7d ticks" % (nm, np.median(ph[:, i]), np.percentile(ph[:, i], 10),
                                                                np.percentile(ph[:, i], 90)))
