#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max, share) of a rocprofv3 rocpd database.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--csv out.csv] [--grid]

rocprofv3 --kernel-trace --stats writes an SQLite file in this image; this prints the table the
CSV "kernel_stats" output would hold.  With --grid the rows are split by launch grid as well.
"""
import argparse
import re
import sqlite3


def short(name):
    m = re.match(r"(?:void )?(?:ptd::)?([A-Za-z_0-9]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--grid", action="store_true")
    ap.add_argument("--match", help="only kernels whose name contains this")
    ap.add_argument("--last", type=int, help="only the last N calls of every row (e.g. the final pass of a run)")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    sel = "name, (end - start)" + (", 'grid ' || %s || 'x' || grid_y" % gcol if (a.grid and gcol) else "")
    rows = {}
    for r in db.execute("select %s from kernels order by start" % sel):
        if a.match and a.match not in r[0]:
            continue
        key = (short(r[0]),) + tuple(r[2:])
        rows.setdefault(key, []).append(r[1])
    if a.last:
        rows = {k: v[-a.last:] for k, v in rows.items()}
    total = sum(sum(v) for v in rows.values())
    out = []
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        out.append((" ".join(str(x) for x in k), len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / total))
    print("%-44s %8s %14s %12s %10s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for o in out:
        print("%-44s %8d %14d %12.0f %10d %12d %7.2f" % o)
    if a.csv:
        with open(a.csv, "w") as f:
            f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
            for o in out:
                f.write('"%s",%d,%d,%.1f,%d,%d,%.3f\n' % o)


if __name__ == "__main__":
    main()
