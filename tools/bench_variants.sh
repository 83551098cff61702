#!/bin/bash
# Bench a list of library builds on the box: tools/bench_variants.sh [bench args] -- lib1.so lib2.so ...
# Every run has its own time limit and the script stops at the first failure.
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
for v in "$@"; do
    echo "== $v"
    MI_PATHTRACE_LIB=$v timeout -k 10 200 python bench.py "${args[@]}" --no-cpu-baseline 2>/dev/null > /tmp/bv.json || exit 1
    python3 -c "
import json
d = json.loads(open('/tmp/bv.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['ms_per_spp'], d['roofline']['stage_ms_per_spp'])" || exit 1
done
