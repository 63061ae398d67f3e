#!/usr/bin/env python3
"""Standalone timing of the implicit-GEMM kernel through dv_op_linear (GPU box), for PMC runs."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diff_vits_amd  # noqa
from diff_vits_amd import _lib as L

shapes = [(8192, 128, 128), (4096, 256, 2048), (1024, 3072, 512), (2048, 384, 3072), (2048, 1536, 384)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for M, K, N in shapes:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    for _ in range(3):
        L.check(L.lib().dv_op_linear(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), M, K, N, 0, None))
    torch.cuda.synchronize()
    ref = x @ w.t() + b
    print("M=%d K=%d N=%d rel err %.2e" % (M, K, N, float((y - ref).norm() / ref.norm())))
