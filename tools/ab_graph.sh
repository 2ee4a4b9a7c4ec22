#!/bin/bash
# A/B inside ONE job on the GPU box (boxes differ by up to 10 %): alternating graph-replayed bench runs under two environments.
#   bash tools/ab_graph.sh "SPV_TN_DEPTH=3" "SPV_TN_DEPTH=1" [rounds] [extra bench args]
# The switches exist only in the lab build (make -C vit-spectre-experiments_amd/csrc lab) and in the host's lab mode (SPV_LAB=1);
# "SPV_LIB_PATH=/path/to/other/libspv_hip.so" compares two builds of the library instead.
A="$1"; B="$2"; R="${3:-3}"; shift 3 2>/dev/null
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LABLIB="$ROOT/vit-spectre-experiments_amd/lib/libspv_hip_lab.so"
for i in $(seq 1 $R); do
  for E in "$A" "$B"; do
    base="SPV_LAB=1"
    if [ -f "$LABLIB" ] && [[ "$E" != *SPV_LIB_PATH* ]]; then base="$base SPV_LIB_PATH=$LABLIB"; fi
    v=$(env $base $E python3 "$ROOT/bench.py" --no-cpu-baseline --variants none --no-roofline --no-every-row --no-dp-sequence --no-script-leg --steps 40 --warmup 10 "$@" 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['eager']['ms_per_step'])")
    echo "$E : graph/eager ms $v"
  done
done
