#!/bin/bash
# Round-N measurement set for the headline (run on the MI355X box from the repo root):
#   gpurun_out/<tag>/bench_n1.json                 default `python bench.py` line
#   gpurun_out/<tag>/kstats/*_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command (20 steps)
#   gpurun_out/<tag>/pmc/*_counter_collection.csv  separate --pmc FETCH_SIZE pass (never mixed with other tracing)
# The summaries to be judged are then copied into profiles/ by hand (profiles/README.md).
set -eu -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)   # the checkout this script lives in (no harness variable needed)
TAG=${1:-r04}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
python "$ROOT/bench.py" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
tail -c 600 "$OUT/bench_n1.json"; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kstats" -o bench -- python3 "$ROOT/bench.py" --steps 20 --no-cpu-baseline --no-encoder-leg --no-rerank-leg --no-ivf-leg > "$OUT/bench_under_rocprof.json" 2> "$OUT/kstats.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc" -o bench -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --latency-steps 0 --no-cpu-baseline --no-encoder-leg --no-rerank-leg --no-ivf-leg > "$OUT/bench_under_pmc.json" 2> "$OUT/pmc.err"
ls "$OUT/kstats" "$OUT/pmc"
