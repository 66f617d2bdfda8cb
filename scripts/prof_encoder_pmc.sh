#!/bin/bash
# Query encoder (bge-base architecture, 32 queries): kernel summary and MFMA-busy counters (separate --pmc pass, kernel trace only).
set -eu -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r04}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kstats_encoder" -o enc -- python3 "$ROOT/scripts/prof_encoder.py" > "$OUT/kstats_encoder.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_encoder" -o enc -- python3 "$ROOT/scripts/prof_encoder.py" > "$OUT/pmc_encoder.log" 2>&1
ls "$OUT/kstats_encoder" "$OUT/pmc_encoder"
