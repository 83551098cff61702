#!/bin/bash
# All workloads of bench.py, one JSON line each into gpurun_out/bench_<workload>.json; stops at the first failure.
for w in cornell sss sss32 dragon chess halo; do
    echo "== $w"
    timeout -k 10 500 python bench.py --workload $w --no-cpu-baseline "$@" 2>gpurun_out/bench_$w.err > gpurun_out/bench_$w.json || { tail -5 gpurun_out/bench_$w.err; exit 1; }
    python3 -c "
import json
d = json.loads(open('gpurun_out/bench_$w.json').read().strip().splitlines()[-1])
print(d['value'], 'Mpb/s', d['ms_per_step'], 'ms/spp', d['config']['ms_per_spp'], d['roofline']['stage_ms_per_spp'])" || exit 1
done
