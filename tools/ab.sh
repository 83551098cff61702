#!/bin/bash
# A/B of library builds on the box:  tools/ab.sh "workload1 workload2" variant1 variant2 ...
# variant = name of variants/libmi_pathtrace_<name>.so ("base" = the library in csrc), optionally name:VAR=value to set an environment variable
# One bench run per (workload, build), each under its own time limit; stops at the first failure.
W=$1; shift
run() { echo "== $1 $3 $4"; env $4 MI_PATHTRACE_LIB=$2 timeout -k 10 280 python bench.py --workload $3 --steps 3 --no-cpu-baseline --no-extras 2>/tmp/bv.err > /tmp/bv.json || { echo FAILED; tail -5 /tmp/bv.err; exit 1; }
  python3 -c "
import json
d = json.loads(open('/tmp/bv.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['ms_per_spp'], d['roofline']['stage_ms_per_spp'])"; }
for w in $W; do for spec in "$@"; do
  v=${spec%%:*}; e=""; [ "$v" != "$spec" ] && e=${spec#*:}
  if [ "$v" = base ]; then run base cuda-path-tracer_amd/csrc/libmi_pathtrace.so $w $e || exit 1; else run $v variants/libmi_pathtrace_$v.so $w $e || exit 1; fi
done; done
