#!/bin/bash
# Regenerate the judged profile artefacts ON THE GPU BOX (via gpurun):  bash tools/refresh_profiles.sh r03_v1
# Writes gpurun_out/<tag>/: bench lines, rocprofv3 --kernel-trace --stats summaries, an ordered trace of one replayed step, PMC traffic.
set -u
TAG=${1:-r03_v1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
LEGS="--no-cpu-baseline --no-every-row --no-dp-sequence --no-script-leg --no-base224 --variants none"
echo "== default line (what the driver runs)"; python3 "$ROOT/bench.py" > "$OUT/${TAG}_default_bench_line.json" 2> "$OUT/default_bench.err" || echo "default bench failed"
for MIX in fft permut dwt_embed; do
  echo "== bench $MIX"; python3 "$ROOT/bench.py" --mixer $MIX $LEGS > "$OUT/${TAG}_${MIX}_bs512_bench.json" 2> "$OUT/${MIX}_bench.err" || echo "bench $MIX failed"
  echo "== stats $MIX"
  rm -rf /tmp/prof_$MIX
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$MIX -o p -- python3 "$ROOT/bench.py" --mixer $MIX --steps 20 --warmup 5 --no-roofline $LEGS > "$OUT/${MIX}_stats.log" 2>&1 || echo "stats $MIX failed"
  F=$(find /tmp/prof_$MIX -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp "$F" "$OUT/${TAG}_${MIX}_bs512_kernel_stats.csv"
  T=$(find /tmp/prof_$MIX -name "*kernel_trace.csv" | head -1)
  # the bench ends with the eager leg (>= 7 steps): step 12 from the end is a graph replay
  [ -n "$T" ] && python3 "$ROOT/tools/step_trace.py" "$T" 12 > "$OUT/${TAG}_${MIX}_step_trace_graph.txt"
done
echo "== dp sequence"; python3 "$ROOT/bench.py" --dp-sequence $LEGS > "$OUT/${TAG}_fft_dp_sequence_bench.json" 2> "$OUT/dp_bench.err" || echo "dp-sequence bench failed"
for MIX in fft permut; do
  echo "== pmc $MIX"
  python3 "$ROOT/tools/collect_pmc.py" ${TAG}_${MIX} --mixer $MIX > "$OUT/${MIX}_pmc.log" 2>&1 || echo "pmc $MIX failed"
  [ -f "$ROOT/gpurun_out/${TAG}_${MIX}_pmc_traffic.json" ] && cp "$ROOT/gpurun_out/${TAG}_${MIX}_pmc_traffic.json" "$OUT/"
done
echo "== vit"; python3 "$ROOT/bench.py" --model vit --steps 10 --warmup 3 --no-roofline > "$OUT/${TAG}_vit_bs512_bench.json" 2> "$OUT/vit_bench.err" || echo "vit failed"
echo "== infer"; python3 "$ROOT/tools/infer_bench.py" --mixer fft > "$OUT/${TAG}_infer_fft.json" 2>/dev/null
python3 "$ROOT/tools/infer_bench.py" --mixer permut > "$OUT/${TAG}_infer_permut.json" 2>/dev/null
ls -la "$OUT"
