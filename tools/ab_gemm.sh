R=$GRAFT_REPO_ROOT
python3 -m pytest $R/tests/test_gpu_ops.py -x -q -k "gemm" 2>&1 | tail -1
python3 $R/tools/strip_acc_bench.py 2>&1 | grep -v amdgpu
