#!/bin/bash
# hardware counters of the default mode's two-plane GEMM (gemm_nt_wl_kernel<2,1,3>) on one shape; bounded passes
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r04p}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 5 110 rocprofv3 --pmc $line --kernel-trace --output-format csv -d "$OUT/q$i" -o rr -- "$ROOT/scripts/exp/gemm_wh_bench" 178405 128 > "$OUT/q$i.log" 2>&1
  echo "pass $i rc=$? : $line"
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum
TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum
GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
SETS
