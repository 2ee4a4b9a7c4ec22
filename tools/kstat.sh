R=$GRAFT_REPO_ROOT
python3 -m pytest $R/tests -m gpu -x -q 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_m
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > /dev/null 2>&1
F=$(find /tmp/prof_m -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms/step", tot/25/1e6)
for r in rows[:12]:
    print(f'{r["Name"][:72]:72s} {int(r["Calls"])/25:6.1f}/step  avg {float(r["AverageNs"])/1e3:7.1f} us')
PY
for i in 1 2 3; do python3 $R/bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'; done
