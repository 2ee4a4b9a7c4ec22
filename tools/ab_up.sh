R=$GRAFT_REPO_ROOT
# (lab switches: needs `make -C vit-spectre-experiments_amd/csrc lab`, SPV_LAB=1 and SPV_LIB_PATH=.../lib/libspv_hip_lab.so in the environment)
python3 -m pytest $R/tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do for S in 0 1; do echo "no_up=$S $(if [ $S = 1 ]; then export SPV_TAIL_NO_UP=1; else unset SPV_TAIL_NO_UP; fi; python3 $R/bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["final_loss"])')"; done; done
