#!/usr/bin/env python3
"""GPU box: per-operation time inside the persistent launch (s_memtime of one workgroup) next to the per-launch times."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["DVITS_PERSIST"] = "1"
import numpy as np
import torch
import diff_vits_amd  # noqa
from diff_vits_amd import synth, _lib as L
import persist_check as pc

B, T, Lp = 8, 1024, 256
m = pc.build()
x = torch.from_numpy(synth.normal(1, "x", (B, 80, T))).cuda(); cond = torch.from_numpy(synth.normal(1, "c", (B, 128, T))).cuda()
enc = torch.from_numpy(synth.normal(1, "e", (B, Lp, 128))).cuda(); t = torch.full((B,), 500.0, device="cuda")
eng = m.hip_engine(); eng.sync_weights(); eng.prepare(B, T, Lp); eng.set_cond(enc, None)
for _ in range(3):
    eng.eval(x, cond, t)
torch.cuda.synchronize()
ticks = (C.c_uint64 * 1024)(); first = C.c_int32()
n = L.lib().dv_unet_persist_ticks(eng.handle, C.byref(first), ticks, 1024)
tk = np.array(ticks[:n], dtype=np.int64)
rows = eng.profile_forward(x, cond, t)          # per-launch schedule, event pair per op
rows = [eng.profile_forward(x, cond, t) for _ in range(3)][-1]
d = np.diff(tk)
tot = d.sum()
print("persistent launch: %d ops, %d ticks total (~%.2f ms at 1.8 GHz)" % (n - 1, tot, tot / 1.8e6))
agg = {}
for i in range(n - 1):
    kind, fl, ms, desc = rows[first.value + i]
    a = agg.setdefault((kind, desc), [0, 0, 0.0])
    a[0] += 1; a[1] += d[i]; a[2] += ms * 1e3
print("%-10s %-62s %5s %12s %12s" % ("kind", "shape", "n", "persist us", "per-launch us"))
for (kind, desc), (cnt, tks, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-10s %-62s %5d %12.1f %12.1f" % (kind, desc[:62], cnt, tks / 1800.0 / cnt, us / cnt))
bykind = {}
for (kind, desc), (cnt, tks, us) in agg.items():
    b = bykind.setdefault(kind, [0, 0.0, 0.0]); b[0] += cnt; b[1] += tks / 1800.0; b[2] += us
for k, (cnt, pus, lus) in bykind.items():
    print("TOTAL %-8s n=%3d  persistent %.3f ms   per-launch (event pairs) %.3f ms" % (k, cnt, pus / 1e3, lus / 1e3))
