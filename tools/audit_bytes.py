#!/usr/bin/env python3
"""L2-miss bytes per path-bounce of k_shade / k_prims for every variant run by tools/audit_bytes.sh.

    python tools/audit_bytes.py gpurun_out/audit_<workload> [out.txt]

Sums the L2's own 32-byte tallies (TCC_EA0_RDREQ_DRAM_32B, TCC_EA0_WRREQ_WRITE_DRAM_32B) over the dispatches of the
last, single-stream pass of bench.py, as tools/pmc_workload.py does.  A variant's paths are not the base's (its
lookups return other values), so every figure is divided by that run's own path-bounce count.
"""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_workload import load, last_sum  # noqa: E402


def main():
    base = sys.argv[1]
    lines = []
    for bj in sorted(glob.glob(os.path.join(base, "*_bench.json"))):
        v = os.path.basename(bj)[:-len("_bench.json")]
        bench = json.loads(open(bj).read().strip().splitlines()[-1])
        n = bench["steps"] * bench["config"]["depth"]
        pb = bench["config"]["path_bounces_per_spp"] * bench["config"]["spp_per_step_per_gpu"] * bench["steps"]
        d = load(os.path.join(base, v))
        for k in ("k_shade", "k_prims"):
            if k not in d:
                continue
            s = last_sum(d[k], n)
            rd, wr = s.get("TCC_EA0_RDREQ_DRAM_32B", 0.0) * 32.0, s.get("TCC_EA0_WRREQ_WRITE_DRAM_32B", 0.0) * 32.0
            lines.append("%-10s %-8s read %7.1f B  written %6.1f B per path-bounce   (%d launches, %.3e path-bounces, bench value %.0f)" % (
                v, k, rd / pb, wr / pb, s["_launches"], pb, bench["value"]))
    text = "\n".join(lines) + "\n"
    print(text, end="")
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)


if __name__ == "__main__":
    main()
