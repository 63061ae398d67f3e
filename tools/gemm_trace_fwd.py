#!/usr/bin/env python3
"""Phase timeline of EVERY implicit-GEMM launch of one denoiser forward at the bench shape (GPU box, development build:
`make -C diff-vits_amd/csrc trace`).  For launch n = 0, 1, ... the trace library stamps only that launch (dv_debug_gemm_trace_select),
one eager forward runs, and the medians over its workgroups are printed: prologue / first tile / k-loop / k-group reduction /
split-K pair hand-over / epilogue split (bias+residual -> stores -> statistics -> in-epilogue GroupNorm wait -> apply)."""
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
import bench  # noqa: E402
from diff_vits_amd import synth  # noqa: E402

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev, "bf16x3")
x, cond, enc, mask = (torch.from_numpy(v).to(dev) for v in synth.make_inputs(B, 80, T, 256))
eng = model.hip_engine()
eng.prepare(B, T, 256)
eng.set_cond(enc, None)
t = torch.full((B,), 500.0, device=dev)
lib.dv_debug_gemm_trace.restype = C.c_int
lib.dv_debug_gemm_trace.argtypes = [C.c_void_p, C.c_int]
lib.dv_debug_gemm_trace_select.argtypes = [C.c_int]
lib.dv_debug_gemm_trace_desc.restype = C.c_int
lib.dv_debug_gemm_trace_desc.argtypes = [C.c_char_p, C.c_int]
W, NWG = 32, 8192
for _ in range(2):
    eng.eval(x, cond, t)
torch.cuda.synchronize()
lib.dv_debug_gemm_trace_select(1 << 30)
eng.eval(x, cond, t)
torch.cuda.synchronize()
dbuf = C.create_string_buffer(256)
n_gemm = lib.dv_debug_gemm_trace_desc(dbuf, 256)
print("%d GEMM launches per forward (B=%d, T=%d); cycles = s_memtime ticks, medians over the launch's workgroups" % (n_gemm, B, T))
# (differences are taken inside one workgroup only: s_memtime counters of different XCDs are not aligned)
print("  n  wgs | prologue  lands  k-loop  kgrp-add  pair | bias+res  stores  stats  gnx-wait  gnx-sync  rest | whole | what")
buf = np.zeros((NWG, W), dtype=np.uint64)
tot = np.zeros(12)
for n in range(n_gemm):
    lib.dv_debug_gemm_trace_select(n)
    assert lib.dv_debug_gemm_trace_clear() == 0
    eng.eval(x, cond, t)
    torch.cuda.synchronize()
    assert lib.dv_debug_gemm_trace(buf.ctypes.data_as(C.c_void_p), NWG) == 0
    lib.dv_debug_gemm_trace_desc(dbuf, 256)
    tt = buf.astype(np.int64)
    live = (tt[:, 0] > 0)
    fin = live & (tt[:, 5] > 0)          # workgroups that ran an epilogue (a split-K pair's first arriver leaves early)
    a = tt[live]
    f = tt[fin]
    if len(f) == 0:
        print("%3d  (no finishing workgroup stamped) %s" % (n, dbuf.value.decode()))
        continue
    med = lambda v: int(np.median(v)) if len(v) else 0
    d = lambda b, c, src=f: med(src[:, b] - src[:, c])
    has = lambda k: (f[:, k] > 0).all()
    pro, lands, kloop = d(1, 0, a), d(2, 1, a), d(3, 2, a)
    kadd = d(22, 3) if has(22) else 0
    pair = d(4, 22) if has(22) else d(4, 3)
    e16 = d(16, 4) if has(16) else 0
    e17 = d(17, 16) if has(17) else 0
    e18 = d(18, 17) if has(18) and has(17) else 0
    gw = d(20, 19) if has(20) else 0
    gs = d(21, 20) if has(21) else 0
    last = 21 if has(21) else (18 if has(18) else 4)
    rest = d(5, last)
    whole = d(5, 0)
    if "--prologue" in sys.argv:       # where the prologue goes: arguments arrive | row geometry | side DMAs issued | first tile issued
        ok = lambda k: (a[:, k] > 0).all()
        parts = [d(8, 0, a) if ok(8) else 0, d(9, 8, a) if ok(9) and ok(8) else 0, d(10, 9, a) if ok(10) and ok(9) else 0,
                 d(1, 10, a) if ok(10) else 0]
        print("%3d %4d | args %6d  rows %6d  side %6d  first-tile %6d | prologue %6d | whole %6d | %s"
              % ((n, len(a)) + tuple(parts) + (pro, whole, dbuf.value.decode())))
        continue
    row = [pro, lands, kloop, kadd, pair, e16, e17, e18, gw, gs, rest, whole]
    tot += np.array(row, dtype=np.float64)
    print("%3d %4d | %7d %6d %7d %8d %6d | %7d %7d %6d %8d %8d %6d | %6d | %s"
          % ((n, len(a)) + tuple(row) + (dbuf.value.decode(),)))
print("sum      | %7d %6d %7d %8d %6d | %7d %7d %6d %8d %8d %6d | %6d" % tuple(int(v) for v in tot))
