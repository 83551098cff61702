#!/bin/bash
# Hardware-counter passes over tools/diag/primary_set.py (one closest-hit launch over all camera rays): tools/pmc_ray_set.sh <workload> [out]
set -e -o pipefail
W=${1:-dragon}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/${2:-gpurun_out/rayset_$W}
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -- python3 "$ROOT/tools/diag/primary_set.py" $W > "$OUT/sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq2" -- python3 "$ROOT/tools/diag/primary_set.py" $W > "$OUT/sq2.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr --output-format csv -d "$OUT/l2" -- python3 "$ROOT/tools/diag/primary_set.py" $W > "$OUT/l2.log" 2>&1
echo done
