#!/bin/bash
# Which unit of the CU the kernels keep busy: texture addresser / L1 counters and wave-level instruction counters,
# one pass each, over a short bench.py run.
#   tools/profile_units.sh [out dir under gpurun_out]
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/${1:-gpurun_out/prof_units}
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/ta" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 2 > "$OUT/ta_bench.json"
echo "ta pass done"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/sq" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 2 > "$OUT/sq_bench.json"
echo "sq pass done"
