#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values over the dispatches of kernels whose name contains a pattern."""
import csv
import glob
import sys
from collections import defaultdict

d, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
for k in sorted(acc):
    print("%-28s %14.0f  (n=%d)" % (k, acc[k][0] / acc[k][1], acc[k][1]))
