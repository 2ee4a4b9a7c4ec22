#!/bin/bash
# ON THE GPU BOX (via gpurun):  bash tools/step_check.sh <tag> [mixer] [pytest paths ...]
# A quick look at one replayed step after a kernel change: optional tests, the bench line of one mixer, the kernel stats and the ordered
# trace of one graph replay.  Writes gpurun_out/<tag>_*.
set -u
TAG=${1:-chk}; MIX=${2:-fft}; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
if [ $# -gt 0 ]; then
  (cd "$R" && timeout -k 10 900 python3 -m pytest "$@" -x -q -m gpu > "$R/gpurun_out/${TAG}_tests.txt" 2>&1); rc=$?
  tail -3 "$R/gpurun_out/${TAG}_tests.txt"
  [ $rc -ne 0 ] && exit $rc
fi
cd /tmp
L="--no-cpu-baseline --no-every-row --no-dp-sequence --no-script-leg --no-base224 --variants none"
python3 "$R/bench.py" --mixer $MIX $L > "$R/gpurun_out/${TAG}_${MIX}.json" 2> "$R/gpurun_out/${TAG}_${MIX}.err" || { echo "bench failed"; tail -5 "$R/gpurun_out/${TAG}_${MIX}.err"; exit 1; }
rm -rf /tmp/prof_$MIX
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$MIX -o p -- python3 "$R/bench.py" --mixer $MIX --steps 20 --warmup 5 --no-roofline $L > "$R/gpurun_out/${TAG}_trace.log" 2>&1
F=$(find /tmp/prof_$MIX -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" "$R/gpurun_out/${TAG}_${MIX}_kernel_stats.csv"
T=$(find /tmp/prof_$MIX -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python3 "$R/tools/step_trace.py" "$T" 12 > "$R/gpurun_out/${TAG}_${MIX}_step_trace.txt"
tail -${TAIL:-18} "$R/gpurun_out/${TAG}_${MIX}_step_trace.txt" | cut -c1-130
python3 -c "
import json; d=json.load(open('$R/gpurun_out/${TAG}_${MIX}.json')); r=d['roofline']; print('ms/step', d['ms_per_step'], 'img/s', d['value'], 'dominant', r['kernel'], r['avg_us'], 'us frac', r['frac'])"
