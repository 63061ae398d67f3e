import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DVITS_KEEP_INTERMEDIATES"] = "1"
import numpy as np, torch
import diff_vits_amd
from diff_vits_amd import synth
sys.path.insert(0, os.path.join(ROOT, "tools"))
from persist_check import build

def run(B, T, L, persist, names):
    os.environ["DVITS_PERSIST"] = "1" if persist else "0"
    m = build()
    x = torch.from_numpy(synth.normal(1, "x", (B, 80, T))).cuda(); cond = torch.from_numpy(synth.normal(1, "c", (B, 128, T))).cuda()
    enc = torch.from_numpy(synth.normal(1, "e", (B, L, 128))).cuda(); t = torch.full((B,), 500.0, device="cuda")
    eng = m.hip_engine(); eng.sync_weights(); eng.prepare(B, T, L); eng.set_cond(enc, None)
    y = eng.eval(x, cond, t); torch.cuda.synchronize()
    out = {}
    for n in names:
        try: out[n] = eng.probe(n).numpy()
        except Exception as e: out[n] = None
    return out, eng.persist_status()

names = ["conv_in"]
for i in range(2):
    names += ["down_blocks.0.resnets.%d.conv1" % i, "down_blocks.0.resnets.%d" % i, "down_blocks.0.attentions.%d.proj_in" % i,
              "down_blocks.0.attentions.%d.transformer_blocks.0.attn1" % i, "down_blocks.0.attentions.%d.transformer_blocks.0.attn2" % i,
              "down_blocks.0.attentions.%d.transformer_blocks.0.ff" % i, "down_blocks.0.attentions.%d" % i]
for ub in range(4):
    for i in range(3):
        names += ["up_blocks.%d.resnets.%d.conv1" % (ub, i), "up_blocks.%d.resnets.%d" % (ub, i)]
        if ub > 0:
            names += ["up_blocks.%d.attentions.%d.proj_in" % (ub, i), "up_blocks.%d.attentions.%d.transformer_blocks.0.attn1" % (ub, i),
                      "up_blocks.%d.attentions.%d.transformer_blocks.0.attn2" % (ub, i), "up_blocks.%d.attentions.%d.transformer_blocks.0.ff" % (ub, i), "up_blocks.%d.attentions.%d" % (ub, i)]
    names += ["up_blocks.%d.upsamplers.0" % ub]
names += ["down_blocks.0.downsamplers.0", "down_blocks.1.resnets.0.conv1", "down_blocks.1.resnets.0", "down_blocks.1.attentions.0", "down_blocks.2.attentions.1", "mid_block.attentions.0", "up_blocks.0.resnets.0", "up_blocks.3.attentions.2"]
B, T, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
a, _ = run(B, T, L, False, names)
b, st = run(B, T, L, True, names)
print("persist status", st)
for n in names:
    if a[n] is None or b[n] is None: print("%-70s missing" % n); continue
    d = np.linalg.norm(a[n].astype(np.float64) - b[n]) / max(1e-30, np.linalg.norm(a[n].astype(np.float64)))
    per_b = [float(np.linalg.norm(a[n][i].astype(np.float64) - b[n][i]) / max(1e-30, np.linalg.norm(a[n][i].astype(np.float64)))) for i in range(B)]
    if d > 1e-7: print("%-70s rel %.2e   per item: %s" % (n, d, " ".join("%.1e" % v for v in per_b)))
