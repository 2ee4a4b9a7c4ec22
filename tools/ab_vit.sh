R=$GRAFT_REPO_ROOT
# (lab switches: needs `make -C vit-spectre-experiments_amd/csrc lab`, SPV_LAB=1 and SPV_LIB_PATH=.../lib/libspv_hip_lab.so in the environment)
python3 -m pytest $R/tests/test_gpu_ops.py -x -q -k "gemm_nt" 2>&1 | tail -1
for i in 1 2; do for S in 1 0; do echo "vit strip=$S $(SPV_GEMM_STRIP=$S python3 $R/bench.py --model vit --steps 10 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"; done; done
