#!/bin/bash
# Profiles of the default bench.py run on the GPU box: kernel trace, then the two HBM counter passes (each its own run).
#   tools/profile_round.sh [out dir under gpurun_out]      then, here:
#   python tools/rocpd_stats.py <out>/kt/*/*.db --grid ; python tools/pmc_traffic.py <out>/fetch <out>/write 64 profiles/pmc_traffic.json
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/${1:-gpurun_out/prof}
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/kt_bench.json"
echo "kernel trace done"
timeout -s KILL 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 2 > "$OUT/fetch_bench.json"
echo "fetch pass done"
timeout -s KILL 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 2 > "$OUT/write_bench.json"
echo "write pass done"
