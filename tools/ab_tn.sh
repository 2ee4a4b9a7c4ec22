R=$GRAFT_REPO_ROOT
# (lab switches: needs `make -C vit-spectre-experiments_amd/csrc lab`, SPV_LAB=1 and SPV_LIB_PATH=.../lib/libspv_hip_lab.so in the environment)
for i in 1 2 3; do for D in 3 1; do echo "tn depth=$D $(SPV_TN_DEPTH=$D python3 $R/bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"; done; done
for D in 3 1; do SPV_TN_DEPTH=$D python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('depth $D:', [(k['shape'], k['avg_us']) for k in d['kernels'][:4]])"; done
