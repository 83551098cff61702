#!/bin/bash
# Kernel trace of the default bench.py run only (see tools/profile_round.sh for the counter passes).
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/${1:-gpurun_out/prof_kt}
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/kt_bench.json"
