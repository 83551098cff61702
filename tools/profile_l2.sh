#!/bin/bash
# L2 hit / miss counts per kernel for the default bench.py run (one counter pass of its own).
#   tools/profile_l2.sh [out dir under gpurun_out]
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/${1:-gpurun_out/prof_l2}
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/l2" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 2 > "$OUT/l2_bench.json"
echo "l2 pass done"
