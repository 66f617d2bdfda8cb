#!/bin/bash
# MFMA-busy counters of the cross-encoder's GEMM kernels (separate --pmc pass, kernel trace only).
set -eu -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r03}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_rerank" -o rr -- python3 "$ROOT/scripts/prof_rerank.py" > "$OUT/pmc_rerank.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kstats_rerank" -o rr -- python3 "$ROOT/scripts/prof_rerank.py" > "$OUT/kstats_rerank.log" 2>&1
ls "$OUT/pmc_rerank" "$OUT/kstats_rerank"
