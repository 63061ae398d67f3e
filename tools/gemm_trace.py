#!/usr/bin/env python3
"""Per-workgroup phase timeline of the implicit-GEMM kernel (GPU box, development build).

Needs `make -C diff-vits_amd/csrc trace` (libdvits_hip_trace.so, compiled with -DDV_GEMM_TRACE: thread 0 of
each workgroup stamps s_memtime at entry / prologue issued / first tile landed / k-loop done / k-split reduced /
end).  Prints the phase medians per shape and how the workgroups were spread over time."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import diff_vits_amd  # noqa
from diff_vits_amd import _lib as L

L.LIB_PATH = os.environ.get("DVITS_TRACE_LIB") or os.path.join(os.path.dirname(L.LIB_PATH), "libdvits_hip_trace.so")
lib = L.lib()
lib.dv_debug_gemm_trace.restype = C.c_int
lib.dv_debug_gemm_trace.argtypes = [C.c_void_p, C.c_int]

shapes = [(8192, 128, 128), (4096, 256, 2048), (1024, 3072, 512), (2048, 384, 3072), (2048, 1536, 384),
          (4096, 256, 256), (2048, 384, 384)]
# "conv:BxTxCinxCoutxk" traces a conv1d (dv_op_conv1d)
conv_shapes = [tuple(int(v) for v in a[5:].split("x")) for a in sys.argv[1:] if a.startswith("conv:")]
plain = [a for a in sys.argv[1:] if not a.startswith("conv:")]
if plain:
    shapes = [tuple(int(v) for v in a.split("x")) for a in plain]
elif conv_shapes:
    shapes = []
NWG = 8192
buf = np.zeros((NWG, 32), dtype=np.uint64)      # DV_TR_W stamps per workgroup (gemm_tile.h)
for spec in shapes + [("conv",) + a for a in conv_shapes]:
    if spec[0] == "conv":
        _, Bn, Tn, Ci, Co, kk = spec
        M, K, N = Bn * Tn, Ci * kk, Co
        x = torch.randn(Bn, Ci, Tn, device="cuda")
        w = torch.randn(Co, Ci, kk, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        y = torch.empty(Bn, Co, Tn, device="cuda")

        def run():
            L.check(lib.dv_op_conv1d(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), Bn, Ci, Tn, Co, kk, 1, 0, 0, None))
    else:
        M, K, N = spec
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        y = torch.empty(M, N, device="cuda")

        def run():
            L.check(lib.dv_op_linear(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), M, K, N, 0, None))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    buf[:] = 0
    assert lib.dv_debug_gemm_trace_clear() == 0
    # the split / statistics / packing kernels of the op run first; the trace only sees the GEMM
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    assert lib.dv_debug_gemm_trace(buf.ctypes.data_as(C.c_void_p), NWG) == 0
    t = buf.astype(np.int64)
    live = t[:, 5] > 0
    n = int(live.sum())
    t = t[live]
    pro = np.stack([t[:, 8] - t[:, 0], t[:, 9] - t[:, 8], t[:, 10] - t[:, 9], t[:, 11] - t[:, 10], t[:, 1] - t[:, 11]], 1)
    print("   k-loop sums of thread 0 (median cyc): waits for its DMA %d | waits at the barrier %d | multiplies + issues %d"
          % tuple(np.median(t[:, 12:15], 0)))
    if True:
        print("   prologue split (median cyc): kernarg-ready %d | row geometry %d | bias/residual/LN setup %d | issue tile 0 %d | "
              "issue tiles 1.. %d" % tuple(np.median(pro, 0)))
    ph = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4],
                   t[:, 5] - t[:, 0]], 1)
    wall = t[:, 7]
    wall0 = wall.min()
    xcc = (t[:, 6] >> 32) & 0xF
    print("M=%d K=%d N=%d  workgroups=%d  (op incl. split kernel %.1f us)" % (M, K, N, n, e0.elapsed_time(e1) * 1e3))
    names = ["issue prologue", "first tile lands", "k-loop", "k-split reduce", "epilogue", "whole workgroup"]
    for i, nm in enumerate(names):
        v = ph[:, i]
        print("   %-18s median %7d cyc   p10 %7d   p90 %7d" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
    start_us = (wall - wall0) / 100.0     # wall_clock64: 100 MHz
    print("   workgroup start (us after the first): median %.1f  p90 %.1f  max %.1f;  per XCC counts %s"
          % (np.median(start_us), np.percentile(start_us, 90), start_us.max(), np.bincount(xcc, minlength=8).tolist()))
    if start_us.max() > 3.0:     # more than one round of workgroups: compare the first round with the later ones
        late = start_us > 3.0
        for nm, sel in (("first round", ~late), ("later rounds", late)):
            print("   %-12s n=%4d  medians: prologue %d  lands %d  k-loop %d  reduce %d  epilogue %d" % (
                (nm, int(sel.sum())) + tuple(int(np.median(ph[sel, i])) for i in range(5))))
    for x in range(1):           # tick rate of s_memtime against the 100 MHz wall clock, from workgroups of one XCC
        sel = xcc == x
        dw = (wall[sel] - wall[sel].min()) / 100.0
        dm = (t[sel, 0] - t[sel, 0].min()).astype(np.float64)
        if dw.max() > 3.0:
            print("   s_memtime ticks per us (XCC %d): %.1f" % (x, np.polyfit(dw, dm, 1)[0]))
    print("   kernel span by wall clock: %.2f us (first workgroup start -> last start) ; last workgroup lifetime %.0f ticks"
          % (start_us.max(), ph[np.argmax(start_us), 5]))
    # in wall-clock units the whole-workgroup duration, to convert cycles -> us
    print("   s_memtime span of all workgroups: %d cyc" % (t[:, 5].max() - t[:, 0].min()))
