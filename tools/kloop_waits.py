#!/usr/bin/env python3
"""Static check of the GEMM k-loops' wait counts (no GPU needed).

The steady-state k-loop of k_gemm keeps NSTAGE - 1 k-tiles of LDS-DMA in flight across its barrier with hand-counted
`s_waitcnt vmcnt(N)`.  hipcc's own wait insertion knows nothing about the DMAs (inline assembly) - but it DOES insert
`s_waitcnt vmcnt(0)` in front of a vector instruction that overwrites a register a compiler-visible load (the residual
prefetch, issued ahead of the loop on purpose) may still be writing.  Whether that happens is a register-allocation
accident: round 4 added two scalars to the epilogue and one of the two inlined copies of the 64x64x64 tile got three
full drains per k-tile (-10 % end to end, found by a same-box A/B, not by any test).  This tool compiles kernels_gemm.hip
to assembly (or reads a given .s), finds every loop that holds MFMAs and LDS-DMAs, and lists the `vmcnt(0)` waits inside.

    python tools/kloop_waits.py [file.s]        exit code 1 if any steady-state k-loop drains the queue
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diff-vits_amd", "csrc")


def compile_asm(src="kernels_gemm.hip", extra=()):
    out = os.path.join(tempfile.mkdtemp(prefix="kloop_"), "k.s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *extra, os.path.join(CSRC, src), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def kloops(asm_path, sym_prefix="_Z6k_gemm"):
    """-> {kernel symbol: [(loop label, instructions, mfma, dma, drains)]} for loops with MFMAs and LDS-DMAs"""
    txt = open(asm_path).read()
    parts = re.split(r"^(%s\w+):.*\n" % re.escape(sym_prefix), txt, flags=re.M)
    res = collections.OrderedDict()
    for i in range(1, len(parts), 2):
        body = parts[i + 1]
        body = body[:body.index(".Lfunc_end")] if ".Lfunc_end" in body else body
        ins = [l.strip() for l in body.split("\n")]
        ins = [l for l in ins if l and not l.startswith(";") and (not l.startswith(".") or l.startswith(".LBB"))]
        labels = {l.split(":")[0]: k for k, l in enumerate(ins) if l.startswith(".LBB")}
        cand = []
        for k, l in enumerate(ins):
            m = re.match(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if not m or m.group(1) not in labels or labels[m.group(1)] >= k:
                continue
            lo = labels[m.group(1)]
            seg = ins[lo:k + 1]
            mfma = sum(1 for s in seg if s.startswith("v_mfma"))
            dma = sum(1 for s in seg if s.startswith("global_load_lds"))
            if mfma and dma:           # (one-wave tiles have no s_barrier)
                cand.append((lo, k, m.group(1), len(seg), mfma, dma, sum(1 for s in seg if s.startswith("s_waitcnt") and "vmcnt(0)" in s)))
        # the steady-state k-loop = an innermost range (layout ranges of outer / unrelated back edges contain it)
        loops = [c[2:] for c in cand if not any(o is not c and c[0] <= o[0] and o[1] <= c[1] for o in cand)]
        res[parts[i]] = loops
    return res


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else compile_asm()
    bad = 0
    for sym, loops in kloops(path).items():
        inner = [l for l in loops if l[4]]
        tag = "DRAINS" if inner else "ok"
        print("%-6s %s  %s" % (tag, sym, [(l[0], "instr=%d mfma=%d dma=%d vmcnt0=%d" % l[1:]) for l in (inner or loops[:1])]))
        bad += bool(inner)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
