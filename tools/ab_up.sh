R=$GRAFT_REPO_ROOT
python3 -m pytest $R/tests -m gpu -x -q 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_m
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > /dev/null 2>&1
F=$(find /tmp/prof_m -name "*kernel_stats.csv" | head -1); grep "tail_bwd_lc\|strip" "$F" | cut -d, -f1-4 | cut -c1-120
for i in 1 2 3; do for S in 0 1; do echo "no_up=$S $(if [ $S = 1 ]; then export SPV_TAIL_NO_UP=1; else unset SPV_TAIL_NO_UP; fi; python3 $R/bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["final_loss"])')"; done; done
