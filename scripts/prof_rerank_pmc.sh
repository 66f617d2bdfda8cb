#!/bin/bash
# MFMA-busy counters of the cross-encoder's GEMM kernels (separate --pmc pass, kernel trace only).
set -e -o pipefail
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_rerank" -o rr -- python3 "$GRAFT_REPO_ROOT/scripts/prof_rerank.py" > "$OUT/pmc_rerank.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kstats_rerank" -o rr -- python3 "$GRAFT_REPO_ROOT/scripts/prof_rerank.py" > "$OUT/kstats_rerank.log" 2>&1
ls "$OUT/pmc_rerank" "$OUT/kstats_rerank"
