#!/bin/bash
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for b in 1024 3072; do
  export PT_PERSIST_BLOCKS=$b PT_STREAMS=1
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/halo_kt_$b -- python3 $ROOT/bench.py --workload halo --steps 1 --no-cpu-baseline > $ROOT/gpurun_out/halo_kt_$b.json 2> $ROOT/gpurun_out/halo_kt_$b.err
  echo done $b
done
