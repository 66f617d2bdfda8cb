#!/bin/bash
# hardware counters of the T16 GEMM on one shape (separate --pmc passes, kernel trace only; every pass bounded)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r04g}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 5 110 rocprofv3 --pmc $line --kernel-trace --output-format csv -d "$OUT/q$i" -o rr -- "$ROOT/scripts/exp/gemm_wh_bench" 178405 16 > "$OUT/q$i.log" 2>&1
  echo "pass $i rc=$? : $line"
done <<'SETS'
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum
TCC_REQ_sum TCC_TAG_STALL_sum
TCC_BUSY_sum TCC_EA0_RDREQ_sum
GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
SETS
