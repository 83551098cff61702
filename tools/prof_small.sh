#!/bin/bash
# Kernel trace of small launch groups: tools/prof_small.sh <spp per step> <steps> <out dir under gpurun_out>
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$3 -- python3 $ROOT/bench.py --spp-per-step $1 --steps $2 --no-cpu-baseline > $ROOT/gpurun_out/$3.json 2> $ROOT/gpurun_out/$3.err
