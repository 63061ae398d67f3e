#!/usr/bin/env python3
"""HBM-side traffic per GEMM launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db > profiles/r01_gemm_hbm_traffic.json

Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes
(MI355X_MICROARCH.md, HBM section), so the read side is doubled; Infinity-Cache hits are included in both."""
import json
import sqlite3
import sys


def per_kernel(db_path, counter, like):
    db = sqlite3.connect(db_path)
    rows = db.execute("select dispatch_id, sum(counter_value) from pmc_events where counter_name = ? and name like ? "
                      "group by dispatch_id", (counter, like)).fetchall()
    vals = [v for _, v in rows]
    return len(vals), (sum(vals) / max(1, len(vals)))


def main():
    fetch_db, write_db = sys.argv[1], sys.argv[2]
    out = {}
    for fam, like in (("gemm", "%k_gemm%"), ("attention", "%k_attention%"), ("gn_apply", "%k_gn_apply%")):
        nf, f = per_kernel(fetch_db, "FETCH_SIZE", like)
        nw, w = per_kernel(write_db, "WRITE_SIZE", like)
        out[fam] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_kib_raw_avg": f, "write_kib_avg": w,
                    "read_bytes_per_launch": f * 2 * 1024, "write_bytes_per_launch": w * 1024,
                    "hbm_bytes_per_launch": (f * 2 + w) * 1024}
    out["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 1 --warmup 0; KiB units; "
                   "gfx950 correction: FETCH_SIZE doubled; Infinity-Cache hits are counted, not excluded")
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
