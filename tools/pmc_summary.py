#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, sum of each counter over dispatches."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    print(k, "dispatches", len(calls[k]))
    for c, v in sorted(acc[k].items()):
        print("    %-32s %.6g" % (c, v))
