#!/bin/bash
# Where the shade stage's L2-miss bytes come from: the request-size counter pass of tools/profile_r03.sh on library variants
# that take one stream of loads away each (built with tools/build_variant.sh aNOTEX -DPT_AUDIT_NOTEX, ... ; "base" = the library).
#   tools/audit_bytes.sh <workload> base aNOTEX aNOCDF aNOENV        then, here:  python tools/audit_bytes.py gpurun_out/audit_<workload>
set -e -o pipefail
W=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/audit_$W
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
ARGS="--workload $W --no-cpu-baseline --no-extras --no-collective-at-one --steps 2 --warmup 0"
for v in "$@"; do
  if [ "$v" = base ]; then export MI_PATHTRACE_LIB=$ROOT/cuda-path-tracer_amd/csrc/libmi_pathtrace.so; else export MI_PATHTRACE_LIB=$ROOT/variants/libmi_pathtrace_$v.so; fi
  timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B TCC_EA0_WRREQ_WRITE_DRAM_32B TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_64B --output-format csv -d "$OUT/$v" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/${v}_bench.json"
  echo "$v done"
done
