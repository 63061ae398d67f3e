#!/usr/bin/env python3
"""The figures the documents quote, from a bench line (default: profiles/r06_bench_final.json) and the PMC JSON beside it."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_bench_final.json")
d = json.loads(open(path).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.0f mel-frames/s, %.2f ms per run, %.3f ms per forward, whole forward %.1f TF/s = %.4f of peak" % (
    d["value"], d["ms_per_step"], r["forward"]["ms_in_graph"], r["forward"]["tflops"], r["forward"]["frac_of_peak"]))
print("roofline family %s: frac %.4f live, %s rocprofv3, time share %.3f, traffic %s, hbm %s GB/s, mfma %s" % (
    r["family"], r["frac"], r["frac_rocprofv3"], r["time_share"], r["traffic"], r["hbm_gbps"], r["mfma_util"]))
for k, f in r["families"].items():
    rp = f["rocprofv3"]
    print("  %-6s launches %3d  live %.2f us  rocprofv3 %s us  %.1f GF  %.1f TF/s  frac %.4f / %s  share %.3f  hbm %s  mfma %s" % (
        k, f["launches_per_forward"], f["avg_launch_us"], rp and round(rp["avg_launch_us"], 2), f["gflop_per_forward"], f["achieved"], f["frac"],
        f["frac_rocprofv3"] and round(f["frac_rocprofv3"], 4), f["time_share"], rp and round(rp["hbm_gbps"]), rp and round(rp["mfma_util"], 3)))
e = d["extra"]
print("unet_rel_l2", e.get("unet_rel_l2"))
for k in ("b16", "config4_unipc20_T2048_B1", "bf16_fast_mode"):
    v = e.get(k)
    if v: print(k, round(v["value"]), round(v["ms_per_run"], 2))
print("batch_sweep", {k: round(v["ms_per_forward"], 3) for k, v in e.get("batch_sweep", {}).items()})
print("b1 T300 latency ms", e.get("b1_T300_L150_unipc30_latency_ms"))
print("cpu_baseline", round(d["cpu_baseline"]["value"], 1), d["cpu_baseline"]["cores"], "speedup", round(d.get("speedup_vs_cpu_baseline", 0), 1))
pj = os.path.join(ROOT, "profiles", "r06_pmc_roofline.json")
if os.path.exists(pj):
    p = json.load(open(pj))
    for k, v in p.items():
        if isinstance(v, dict) and "launches" in v:
            print("  pmc %-10s launches %4d  %.1f us  %.1f MB  %d GB/s  mfma %.3f" % (k, v["launches"], v["avg_us_kernel_trace"], v["hbm_bytes_per_launch"] / 1e6, v["hbm_gbps"], v["mfma_util"]))
    print("  build", p["build"])
