#!/usr/bin/env python3
"""Where a B = 1 sampler run spends its time: busy (kernel durations) vs idle (gaps between dependent kernels) from a
rocprofv3 kernel trace of tools/b1_latency.py.
Usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/b1tr -- python3 $R/tools/b1_latency.py 300
                  python3 tools/b1_timeline.py gpurun_out/b1tr [launches per forward] [forwards of the last run to cover: default 20, 30 = the whole 30-step run]"""
import csv, glob, os, sys, collections

d = sys.argv[1]
per_fwd = int(sys.argv[2]) if len(sys.argv) > 2 else 146
n_fwd = int(sys.argv[3]) if len(sys.argv) > 3 else 20
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last complete sampler run: 30 forwards (+ a few sampler-update kernels); take the last 20 forwards' worth of launches
tail = rows[-n_fwd * (per_fwd + 2):]
span = tail[-1][1] - tail[0][0]
busy = sum(e - s for s, e, _ in tail)
gaps = [max(0, tail[i + 1][0] - tail[i][1]) for i in range(len(tail) - 1)]
print("kernels %d  span %.3f ms  busy %.3f ms (%.1f %%)  idle %.3f ms  avg duration %.2f us  avg gap %.2f us"
      % (len(tail), span / 1e6, busy / 1e6, 100.0 * busy / span, sum(gaps) / 1e6, busy / len(tail) / 1e3, sum(gaps) / len(gaps) / 1e3))
by = collections.defaultdict(lambda: [0, 0, 0])
for i, (s, e, n) in enumerate(tail):
    k = n.split("(")[0][:90]
    by[k][0] += 1
    by[k][1] += e - s
    if i > 0:
        by[k][2] += max(0, s - tail[i - 1][1])
print("%-92s %6s %9s %9s %9s" % ("kernel", "n", "avg us", "gap us", "total ms"))
for k, (n, t, g) in sorted(by.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%-92s %6d %9.2f %9.2f %9.3f" % (k, n, t / n / 1e3, g / n / 1e3, (t + g) / 1e6))
