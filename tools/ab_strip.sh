cd /tmp; export TMPDIR=/tmp
# (lab switches: needs `make -C vit-spectre-experiments_amd/csrc lab`, SPV_LAB=1 and SPV_LIB_PATH=.../lib/libspv_hip_lab.so in the environment)
R=$GRAFT_REPO_ROOT
for S in 1 0; do
  rm -rf /tmp/prof_s$S
  SPV_GEMM_STRIP=$S rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s$S -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > $R/gpurun_out/ab_s$S.log 2>&1
  F=$(find /tmp/prof_s$S -name "*kernel_stats.csv" | head -1); cp "$F" $R/gpurun_out/ab_s${S}_kernel_stats.csv
done
for i in 1 2 3; do for S in 1 0; do for SS in 1 0; do echo "strip=$S side=$SS $(SPV_GEMM_STRIP=$S SPV_SIDE_STREAM=$SS python3 $R/bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"; done; done; done
