R=$GRAFT_REPO_ROOT
# (lab switches: needs `make -C vit-spectre-experiments_amd/csrc lab`, SPV_LAB=1 and SPV_LIB_PATH=.../lib/libspv_hip_lab.so in the environment)
python3 -m pytest $R/tests/test_gpu_ops.py -x -q -k "gemm" 2>&1 | tail -1
python3 $R/tools/strip_acc_bench.py 2>&1 | grep -v amdgpu
