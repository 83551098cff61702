#!/bin/bash
# All workloads of bench.py, one JSON line each into gpurun_out/bench_<workload>.json; stops at the first failure.
for w in cornell sss sss32 duck dragon chess halo p911; do
    echo "== $w"
    timeout -k 10 600 python bench.py --workload $w --no-cpu-baseline --steps 4 "$@" 2>gpurun_out/bench_$w.err > gpurun_out/bench_$w.json || { tail -5 gpurun_out/bench_$w.err; exit 1; }
    python3 -c "
import json
d = json.loads(open('gpurun_out/bench_$w.json').read().strip().splitlines()[-1])
r = d['roofline']
print(d['value'], 'Mpaths*bounce/s', d['ms_per_step'], 'ms/step', d['config']['ms_per_spp'], 'ms/spp', d['config']['iterations_per_launch_group'], 'iterations/group')
print('   stages ms/spp', r['stage_ms_per_spp'])
print('   roofline', r['bound'], r['kernel'], 'achieved', r['achieved'], r['unit'], 'frac', r['frac'], 'hbm_frac', r.get('hbm_frac'), 'boxes/ray', r.get('boxes_per_ray'), 'tris/ray', r.get('tris_per_ray'))
print('   single iteration', d.get('single_iteration'))
print('   ray sets', d.get('ray_sets'))" || exit 1
done
