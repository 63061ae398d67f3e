import os, sys, torch, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import UNET_CASES
from diff_vits_amd import synth
from diff_vits_amd.unet1d.unet_1d_condition import UNet1DConditionModel
kw = UNET_CASES["cfg1"][0]
with torch.device("meta"):
    shapes = {k: tuple(v.shape) for k, v in UNet1DConditionModel(**kw).state_dict().items()}
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(shapes, seed=4321).items()}
B, T, L = (int(v) for v in os.environ.get("SHAPE", "2,2048,300").split(","))
x = torch.from_numpy(synth.normal(11, "x", (B, 80, T))).cuda()
cond = torch.from_numpy(synth.normal(11, "c", (B, 128, T))).cuda()
enc = torch.from_numpy(synth.normal(11, "e", (B, L, 128))).cuda()
mask = torch.ones(B, L, dtype=torch.bool); mask[0, L // 2:] = False; mask = mask.cuda()
t = torch.full((B,), 123.0, device="cuda")
knobsets = {"plain": ("DVITS_GNX", "DVITS_ATTN_FRAG", "DVITS_CHAIN_SPLIT", "DVITS_CHAIN_FF", "DVITS_XCD_N", "DVITS_CHAIN", "DVITS_STAT16"), "fused": ()}
# usage: SHAPE=B,T,L python tools/flaky_repeat.py [plain|fused|<ENV_KNOB>] ...  (one mode per process is safest: some
# knobs are read once per process)
for name in sys.argv[1:] or ["plain", "fused"]:
    ks = knobsets.get(name, (name,))
    for k in ks: os.environ[k] = "0"
    m = UNet1DConditionModel(backend="hip", **kw).eval(); m.load_state_dict(sd); m = m.cuda()
    with torch.no_grad():
        y0 = m(torch.cat([x, cond], 1), t, enc, encoder_attention_mask=mask).sample.clone()
        bad = 0
        for i in range(60):
            # disturb the caches between runs
            junk = torch.randn(64 << 20, device="cuda"); junk.mul_(2.0); del junk
            y = m(torch.cat([x, cond], 1), t, enc, encoder_attention_mask=mask).sample
            if not torch.equal(y, y0):
                bad += 1
                d = (y - y0).abs()
                print(name, "run", i, "differs: max", float(d.max()), "count", int((d > 0).sum()))
    torch.cuda.synchronize()
    print(name, "mismatching runs:", bad, "handover", m.hip_engine().handover_status())
    for k in ks: os.environ.pop(k, None)
