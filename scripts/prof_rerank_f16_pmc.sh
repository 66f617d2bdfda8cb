#!/bin/bash
# RAG_GEMM_F16 cross-encoder pass (fp16 activations in fragment order): per-kernel summary, HBM bytes (FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes, kernel trace only) and MFMA-busy cycles.  Usage: scripts/prof_rerank_f16_pmc.sh [tag] [base]
set -eu -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r03}
WHICH=${2:-}
OUT=$ROOT/gpurun_out/$TAG/f16${WHICH:+_$WHICH}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kstats" -o rr -- python3 "$ROOT/scripts/prof_rerank_f16.py" $WHICH > "$OUT/kstats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o rr -- python3 "$ROOT/scripts/prof_rerank_f16.py" $WHICH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o rr -- python3 "$ROOT/scripts/prof_rerank_f16.py" $WHICH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -o rr -- python3 "$ROOT/scripts/prof_rerank_f16.py" $WHICH > "$OUT/pmc_mfma.log" 2>&1
ls "$OUT"/*
