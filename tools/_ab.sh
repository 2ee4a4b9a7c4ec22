python3 -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for i in 1 2; do for E in "SPV_SIDE_PRIORITY=0" "SPV_SIDE_PRIORITY=1" "SPV_SIDE_PRIORITY=5"; do
 v=$(env $E python3 bench.py --no-cpu-baseline --variants none --no-roofline --steps 40 --warmup 10 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['eager']['ms_per_step'])")
 echo "$E : graph/eager ms $v"; done; done
