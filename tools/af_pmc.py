#!/usr/bin/env python3
"""Run one fused GroupNorm->conv op (dv_op_gn_conv1d) or one plain linear op (dv_op_linear) a few times: the target
of `rocprofv3 --pmc ... -- python3 tools/af_pmc.py af 8 256 384 384 3` / `... lin 2048 1152 384` counter passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diff_vits_amd  # noqa
from diff_vits_amd import _lib as L

lib = L.lib()
kind = sys.argv[1]
a = [int(v) for v in sys.argv[2:]]
if kind == "af":
    Bn, Tn, Ci, Co, kk = a
    x = torch.randn(Bn * Tn, Ci, device="cuda")
    w = torch.randn(Co, Ci, kk, device="cuda") / (Ci * kk) ** 0.5
    b = torch.randn(Co, device="cuda")
    gam, bet = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    y = torch.empty(Bn, Co, Tn, device="cuda")
    for _ in range(10):
        L.check(lib.dv_op_gn_conv1d(L.ptr(x), L.ptr(gam), L.ptr(bet), None, None, L.ptr(w), L.ptr(b), L.ptr(y), Bn, Ci, Tn, Co, kk,
                                    8, 1e-5, 1, 0, None))
else:
    M, K, N = a
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    for _ in range(10):
        L.check(lib.dv_op_linear(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), M, K, N, 0, None))
torch.cuda.synchronize()
