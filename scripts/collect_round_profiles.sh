#!/bin/bash
# Everything the round's profile table quotes, in one pass on the MI355X box (repo root; ~8 min):
#   bench line + kernel stats + FETCH_SIZE pass (collect_bench_profiles.sh), shard step, stages, pipeline configs B / C,
#   2-rank gloo rehearsal of bench.py.  Outputs under gpurun_out/<tag>/; copy the summaries into profiles/.
set -u -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r04}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
bash scripts/collect_bench_profiles.sh "$TAG" > "$OUT/collect.log" 2>&1; echo "bench profiles rc=$?"
python scripts/bench_shard_step.py > "$OUT/shard_step.json" 2> "$OUT/shard_step.err"; echo "shard step rc=$?"
python scripts/bench_stages.py --json "$OUT/stages.json" --no-cpu > "$OUT/stages.txt" 2>&1; echo "stages rc=$?"
python scripts/bench_pipeline.py --profile > "$OUT/pipeline_b.txt" 2>&1; echo "pipeline B rc=$?"
python scripts/bench_pipeline.py --rerank --k 100 --profile > "$OUT/pipeline_c.txt" 2>&1; echo "pipeline C rc=$?"
python scripts/bench_pipeline.py --rerank --k 100 --dtype f16 > "$OUT/pipeline_c_f16.txt" 2>&1; echo "pipeline C f16 rc=$?"
RAG_AMD_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 2 --steps 10 --warmup 3 > "$OUT/bench_n2_gloo.json" 2> "$OUT/bench_n2_gloo.err"; echo "n2 gloo rc=$?"
bash scripts/prof_rerank_pmc.sh "$TAG" > "$OUT/prof_rerank_pmc.log" 2>&1; echo "rerank pmc rc=$?"
bash scripts/prof_rerank_f16_pmc.sh "$TAG" > "$OUT/prof_rerank_f16_pmc.log" 2>&1; echo "rerank f16 pmc rc=$?"
bash scripts/prof_encoder_pmc.sh "$TAG" > "$OUT/prof_encoder_pmc.log" 2>&1; echo "encoder pmc rc=$?"
python scripts/bench_batch_sweep.py > "$OUT/batch_sweep.json" 2> "$OUT/batch_sweep.err"; echo "batch sweep rc=$?"
grep "ms/batch  " "$OUT"/pipeline_*.txt
