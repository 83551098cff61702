#!/usr/bin/env python3
"""HBM traffic per launch of the traversal kernel from two rocprofv3 --pmc passes.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <launches> [out.json]

FETCH_SIZE and WRITE_SIZE are in KiB and need separate passes (TCC counter slots); on gfx950 FETCH_SIZE
tallies 128-byte requests at 64 bytes, so it is doubled.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
The last <launches> dispatches of each kernel are averaged (the timed launch group of bench.py).
"""
import csv
import glob
import json
import sys
from collections import defaultdict

NAMES = {"k_trace<0, false>": "k_trace(closest)", "k_trace<1, false>": "k_trace(shadow)",
         "k_shade<false, false>": "k_shade", "k_prims<false, false>": "k_prims",
         "k_shade<false, true>": "k_shade(sparse form)", "k_prims<false, true>": "k_prims(sparse form)", "k_cull": "k_cull"}


def per_dispatch(root, counter):
    rows = defaultdict(dict)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            for pat, nice in NAMES.items():
                if pat in k:
                    d = rows[nice]
                    d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return rows


def main():
    fdir, wdir, launches = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else None
    fetch, write = per_dispatch(fdir, "FETCH_SIZE"), per_dispatch(wdir, "WRITE_SIZE")
    res, detail = {}, {}
    for k in fetch:
        f = [v for _, v in sorted(fetch[k].items())][-launches:]
        w = [v for _, v in sorted(write.get(k, {}).items())][-launches:]
        fb = 2.0 * 1024.0 * sum(f) / max(len(f), 1)
        wb = 1024.0 * sum(w) / max(len(w), 1)
        res[k] = round(fb + wb)
        detail[k] = {"launches_averaged": len(f), "read_bytes_per_launch(2xFETCH_SIZE)": round(fb), "write_bytes_per_launch": round(wb)}
    res["_detail"] = detail
    res["_method"] = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch, mean of the last %d dispatches" % launches
    print(json.dumps(res, indent=1))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
