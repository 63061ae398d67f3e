#!/usr/bin/env python3
"""Per-kernel-family roofline evidence from rocprofv3 passes over the bench command (GPU box).

Four passes over the SAME command (eager launches: export DVITS_NO_GRAPH=1 first - the counter tool does not survive
hipGraph replay; the program itself directly after `--`):

    cd /tmp && export TMPDIR=/tmp DVITS_NO_GRAPH=1; R=$GRAFT_REPO_ROOT; CMD="python3 $R/bench.py --steps 1 --warmup 1 --solver-steps 6 --no-cpu-baseline --no-roofline"
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_stats -- $CMD
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_fetch -- $CMD
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_write -- $CMD
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/p_mfma -- $CMD
    python3 tools/pmc_roofline.py gpurun_out/p_stats gpurun_out/p_fetch gpurun_out/p_write gpurun_out/p_mfma > profiles/r03_pmc_roofline.json

Units / corrections (MI355X_MICROARCH.md): FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes, so the read side is doubled; Infinity-Cache hits are counted, not excluded.
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16) summed over the chip's 1024 SIMDs;
mfma_util = MFMA busy cycles / (1024 x kernel duration x 2.4 GHz) - the share of the matrix-pipe cycles at the maximum
clock, a lower bound at the actual clock.  (GRBM_GUI_ACTIVE per dispatch reads ~13x the kernel duration in cycles on this
stack - it appears to be summed over the chip's instances - so the ratio against it is reported only as
mfma_util_raw_grbm.)"""
import csv
import glob
import json
import sys
from collections import defaultdict

# (the GEMM family = every launch of the engine's kind "gemm": k_gemm and, since round 5, the resident / streamed convolution kernels)
# (the chain family = the row-block chains and, since round 6, the column-split launches k_qkv_split that replace two of them;
# "qkv_split" lists those on their own as well)
FAMILIES = (("gemm", ("k_gemm", "k_conv3")), ("chain", ("k_chain", "k_qkv_split")), ("ff_split", ("k_ff_split",)), ("attention", ("k_attention",)), ("gn_apply", ("k_gn_apply",)),
            ("conv3", ("k_conv3",)), ("qkv_split", ("k_qkv_split",)))


def counters(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for fam, pats in FAMILIES:
                if any(pat in r["Kernel_Name"] for pat in pats):
                    a = acc[fam][r["Counter_Name"]]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
    return {fam: {c: (v[0] / v[1], v[1]) for c, v in cs.items()} for fam, cs in acc.items()}


def durations(d):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for fam, pats in FAMILIES:
                if any(pat in r["Kernel_Name"] for pat in pats):
                    acc[fam][0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
                    acc[fam][1] += 1
    return {fam: (v[0] / v[1], v[1]) for fam, v in acc.items()}


def main():
    d_stats, d_fetch, d_write, d_mfma = sys.argv[1:5]
    dur, cf, cw, cm = durations(d_stats), counters(d_fetch), counters(d_write), counters(d_mfma)
    out = {}
    for fam, _ in FAMILIES:
        if fam not in dur:
            continue
        us, n = dur[fam]
        f = cf.get(fam, {}).get("FETCH_SIZE", (0.0, 0))[0]
        w = cw.get(fam, {}).get("WRITE_SIZE", (0.0, 0))[0]
        m = cm.get(fam, {})
        busy, act = m.get("SQ_VALU_MFMA_BUSY_CYCLES", (0.0, 0))[0], m.get("GRBM_GUI_ACTIVE", (0.0, 0))[0]
        hbm = (2 * f + w) * 1024
        out[fam] = {"launches": n, "avg_us_kernel_trace": us, "fetch_kib_raw_avg": f, "write_kib_avg": w,
                    "hbm_bytes_per_launch": hbm, "hbm_gbps": hbm / (us * 1e-6) / 1e9 if us else None,
                    "mfma_busy_cycles_per_launch": busy, "sq_busy_cycles_per_launch": m.get("SQ_BUSY_CYCLES", (0.0, 0))[0],
                    "grbm_gui_active_per_launch": act, "mfma_util_raw_grbm": busy / (act * 1024) if act else None,
                    "mfma_util": busy / (1024 * us * 2400.0) if us else None}
    # build identity: bench.py reports these figures only while the loaded library still is this build (dv_version() carries a
    # hash of csrc/); the commit is passed in by the caller (the GPU box has no .git): argv[5]
    try:
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import diff_vits_amd  # noqa: F401
        from diff_vits_amd import _lib
        ver = _lib.lib().dv_version().decode()
    except Exception as e:      # pragma: no cover
        ver = "unknown (%s)" % e
    out["build"] = {"dv_version": ver, "git_head": sys.argv[5] if len(sys.argv) > 5 else None}
    out["note"] = ("rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) and a "
                   "--kernel-trace pass over bench.py, eager launches; FETCH_SIZE doubled (gfx950); Infinity-Cache hits counted; "
                   "mfma_util = MFMA busy cycles / (1024 SIMDs x kernel duration x 2.4 GHz)")
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
