#!/usr/bin/env python3
"""One-line digest of a bench.py JSON line: value, ms per step, ms per spp, per-stage ms per spp."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms/step", d["config"]["ms_per_spp"], "ms/spp", d["roofline"].get("stage_ms_per_spp"))
