#!/bin/bash
# Kernel trace + hardware-counter passes of one bench.py workload on the GPU box (each pass is a run of its own).
#   tools/profile_r03.sh <workload> [out dir under gpurun_out]      (KT_ONLY=1: the kernel trace alone)
# then, here:   python tools/pmc_workload.py gpurun_out/prof_<workload> <workload>
set -e -o pipefail
W=${1:-dragon}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/${2:-gpurun_out/prof_$W}
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
ARGS="--workload $W --no-cpu-baseline --no-extras --no-collective-at-one --steps 2 --warmup 0"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/kt_bench.json"
echo "kernel trace done"
if [ -n "$KT_ONLY" ]; then exit 0; fi
timeout -s KILL 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/sq_bench.json"
echo "sq pass done"
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/sq2_bench.json"
echo "sq2 pass done"
timeout -s KILL 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr --output-format csv -d "$OUT/l2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/l2_bench.json"
echo "l2 pass done"
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/fetch_bench.json"
echo "fetch pass done"
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/write_bench.json"
echo "write pass done"
# request sizes at the L2's memory side: 32-byte units of DRAM reads and writes (a 64-byte request counts 2, a 128-byte one 4), and the
# number of 128- and 64-byte read requests - what FETCH_SIZE's arithmetic has to assume
timeout -s KILL 600 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B TCC_EA0_WRREQ_WRITE_DRAM_32B TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_64B --output-format csv -d "$OUT/ea" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/ea_bench.json"
echo "ea pass done"
