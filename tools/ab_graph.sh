#!/bin/bash
# A/B inside ONE job on the GPU box (boxes differ by up to 10 %): alternating graph-replayed bench runs under two environments.
#   bash tools/ab_graph.sh "SPV_TN_DMA=0" "SPV_TN_DMA=1" [rounds]
A="$1"; B="$2"; R="${3:-3}"
for i in $(seq 1 $R); do
  for E in "$A" "$B"; do
    v=$(env $E python3 bench.py --no-cpu-baseline --variants none --no-roofline --no-every-row --steps 40 --warmup 10 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['eager']['ms_per_step'])")
    echo "$E : graph/eager ms $v"
  done
done
