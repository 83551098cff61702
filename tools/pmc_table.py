"""Per-kernel totals of the counters in a rocprofv3 --pmc CSV.   python tools/pmc_table.py <counter_collection.csv> [name filter]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
dur = collections.defaultdict(dict)
for r in rows:
    k = r["Kernel_Name"].replace("ptd::", "").replace("void ", "")[:34]
    if flt and flt not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
    dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
names = sorted({c for d in agg.values() for c in d})
print("%-34s %5s %9s " % ("kernel", "n", "ms") + " ".join("%14s" % c[-14:] for c in names))
for k, d in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]].values()))[:14]:
    print("%-34s %5d %9.2f " % (k, len(n[k]), sum(dur[k].values())) + " ".join("%14.4g" % d.get(c, 0) for c in names))
