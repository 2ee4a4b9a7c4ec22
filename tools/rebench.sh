# re-run only the bench lines of a profile collection:  bash tools/rebench.sh r01_v17   (on the GPU box)
TAG=${1:-r01_v17}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
for MIX in fft permut dwt_embed; do python3 "$R/bench.py" --mixer $MIX > "$OUT/${TAG}_${MIX}_bs512_bench.json" 2>/dev/null || echo "bench $MIX failed"; done
python3 "$R/bench.py" --model vit --steps 10 --warmup 3 --no-roofline > "$OUT/${TAG}_vit_bs512_bench.json" 2>/dev/null
ls -la "$OUT" | tail -5
