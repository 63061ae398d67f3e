#!/usr/bin/env python3
"""Phase timeline of the column-split block head k_qkv_split (GPU box, `make -C diff-vits_amd/csrc trace`): runs forwards of
the bench model with libdvits_hip_trace.so and prints, for the LAST launch of the selected width, the median s_memtime deltas
between the kernel's phase stamps (thread 0 of every workgroup).   python tools/qkv_trace.py [B T [C]]"""
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
import bench  # noqa: E402
from diff_vits_amd import synth  # noqa: E402

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
CS = [int(sys.argv[3])] if len(sys.argv) > 3 else [128, 256, 384]
dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev, "bf16x3")
x, cond, enc, mask = (torch.from_numpy(v).to(dev) for v in synth.make_inputs(B, 80, T, 256))
eng = model.hip_engine()
eng.prepare(B, T, 256)
eng.set_cond(enc, None)
t = torch.full((B,), 500.0, device=dev)
lib.dv_debug_qkv_trace_select.restype = C.c_int
lib.dv_debug_qkv_trace_select.argtypes = [C.c_int]
lib.dv_debug_qkv_trace.restype = C.c_int
lib.dv_debug_qkv_trace.argtypes = [C.c_void_p, C.c_int]
names = ["args, rows of x + first weights requested", "GroupNorm table, rows -> planes (+ barrier)", "stage 1 k-loop", "quarters summed, h slice written through, flag",
         "wait for the row block's flags", "rows of h back, LayerNorm statistics, planes (+ barrier)", "q: k-loop + epilogue", "k: k-loop + epilogue", "v: k-loop + epilogue"]
for Csel in CS:
    assert lib.dv_debug_qkv_trace_select(Csel) == 0
    for _ in range(3):
        eng.eval(x, cond, t)
    torch.cuda.synchronize()
    NWG = 1024
    buf = np.zeros((NWG, 16), dtype=np.uint64)
    assert lib.dv_debug_qkv_trace(buf.ctypes.data_as(C.c_void_p), NWG) == 0
    tt = buf.astype(np.int64)
    live = tt[:, 9] > 0
    tt = tt[live]
    print("k_qkv_split, C = %d: last launch of the forward, %d workgroups (cycles of s_memtime, thread 0)" % (Csel, int(live.sum())))
    for i, nm in enumerate(names):
        d = tt[:, i + 1] - tt[:, i]
        print("   %-58s median %6d  p10 %6d  p90 %6d" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    for a, b, nm in ((0, 10, "  start -> requests issued"), (10, 11, "  -> GroupNorm entries in LDS (everything requested has landed)"), (11, 12, "  -> table ready"), (12, 1, "  -> rows converted")):
        d = tt[:, b] - tt[:, a]
        print("   %-58s median %6d  p10 %6d  p90 %6d" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    if (tt[:, 13] > 0).all():      # the cross-attention form (MODE 2): stamps 6-9 and 13 mean something else
        for a, b, nm in ((6, 7, "  MODE 2: q pass, query planes, first key tile requested"), (7, 8, "  MODE 2: key loop (+ merge of the key halves)"),
                         (8, 9, "  MODE 2: O stored, flag, wait, O of the row block by DMA"), (9, 13, "  MODE 2: stage 3 (to_out + residual, planes, LN3 partials)")):
            d = tt[:, b] - tt[:, a]
            print("   %-58s median %6d  p10 %6d  p90 %6d" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
        print("   %-58s median %6d" % ("  MODE 2: whole workgroup", np.median(tt[:, 13] - tt[:, 0])))
    print("   %-58s median %6d  max %6d" % ("whole workgroup", np.median(tt[:, 9] - tt[:, 0]), np.max(tt[:, 9] - tt[:, 0])))
    print("   %-58s %6d" % ("first start -> last end over the launch", int(tt[:, 9].max() - tt[:, 0].min())))
