#!/usr/bin/env python3
"""Phase timeline of the row-block chain kernels (GPU box, `make -C diff-vits_amd/csrc trace`): runs one forward of the
bench model with libdvits_hip_trace.so and prints, for the LAST chain launch of the forward (to_out+res+LN+to_q of the last
transformer block, C = 128, M = B*T), the median s_memtime deltas between the kernel's phase stamps."""
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
# optional selection: C amode xa  (e.g. 384 1 0 = the norm+proj_in+LN+q|k|v chain of the 384-channel level)
SEL = [int(v) for v in sys.argv[3:6]] if len(sys.argv) >= 6 else None
dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev, "bf16x3")
x, cond, enc, mask = (torch.from_numpy(v).to(dev) for v in synth.make_inputs(B, 80, T, 256))
eng = model.hip_engine()
eng.prepare(B, T, 256)
eng.set_cond(enc, None)
t = torch.full((B,), 500.0, device=dev)
if SEL:
    lib.dv_debug_chain_trace_select.restype = C.c_int
    lib.dv_debug_chain_trace_select.argtypes = [C.c_int, C.c_int, C.c_int]
    assert lib.dv_debug_chain_trace_select(*SEL) == 0
for _ in range(3):
    eng.eval(x, cond, t)
torch.cuda.synchronize()
NWG = 8192
buf = np.zeros((NWG, 16), dtype=np.uint64)
lib.dv_debug_chain_trace.restype = C.c_int
lib.dv_debug_chain_trace.argtypes = [C.c_void_p, C.c_int]
assert lib.dv_debug_chain_trace(buf.ctypes.data_as(C.c_void_p), NWG) == 0
tt = buf.astype(np.int64)
live = tt[:, 9] > 0
tt = tt[live]
names = ["args+setup -> A requested", "A operand complete (barrier)", "stage-1 k-loop", "hand-over (2 barriers)", "epilogue 1 + LN rows",
         "stage-2 k-loop (pass 0)", "hand-over", "epilogue 2 (pass 0)", "rest"]
print("%s chain launch of the forward: %d workgroups" % ("last selected (C=%d amode=%d xa=%d)" % tuple(SEL) if SEL else "last", int(live.sum())))
for i, nm in enumerate(names):
    d = tt[:, i + 1] - tt[:, i]
    print("   %-32s median %6d cyc  p90 %6d" % (nm, np.median(d), np.percentile(d, 90)))
print("   %-32s median %6d cyc" % ("whole workgroup", np.median(tt[:, 9] - tt[:, 0])))
if (tt[:, 10] > 0).all():       # the cross-attention tail stamps its own phases (thread 0 = head 0)
    xs = [("stage-2 epilogue -> queries in registers", 8, 10), ("cross attention (head 0)", 10, 11), ("wait for the other heads", 11, 12),
          ("stage-3 k-loop", 12, 13), ("hand-over", 13, 14), ("epilogue 3 (+ residual, planes, LN partials)", 14, 9)]
    for nm, a, b in xs:
        d = tt[:, b] - tt[:, a]
        print("   %-44s median %6d cyc  p90 %6d" % (nm, np.median(d), np.percentile(d, 90)))
