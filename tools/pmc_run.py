#!/usr/bin/env python3
"""Per-kernel hardware counters for any python script (development tool; run ON THE GPU BOX via gpurun).

    python tools/pmc_run.py "CTR_A CTR_B,CTR_C CTR_D" [--filter substr] -- tools/gemm_check.py 33280 768 512

Comma-separated groups are separate rocprofv3 --pmc passes (--kernel-trace only, as MI355X_MICROARCH.md prescribes).
Prints the per-launch average of every counter for each kernel whose name contains the filter.
"""
import collections
import csv
import glob
import os
import shutil
import signal
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    sep = args.index("--")
    groups = [g.split() for g in args[0].split(",")]
    flt = args[args.index("--filter") + 1] if "--filter" in args[:sep] else ""
    script = args[sep + 1:]
    script[0] = os.path.join(ROOT, script[0])
    res = collections.defaultdict(dict)
    for gi, g in enumerate(groups):
        out = f"/tmp/pmc_run_{gi}"
        shutil.rmtree(out, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc", *g, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, *script]
        print(f"pass {gi}: {' '.join(g)}", flush=True)
        proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                text=True, start_new_session=True)
        try:
            stdout, _ = proc.communicate(timeout=100)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)  # the process group this tool started, nothing else
            proc.communicate()
            print(f"pass {g} timed out", flush=True)
            continue
        if proc.returncode != 0:
            print(f"pass {g} failed:\n{stdout[-1500:]}", flush=True)
            continue
        f = glob.glob(os.path.join(out, "*", "*counter_collection.csv"))[0]
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in agg.items():
            res[k][c] = (sum(v) / len(v), len(v))
    for k, cs in res.items():
        short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        if flt and flt not in short:
            continue
        print(short)
        for c, (v, n) in sorted(cs.items()):
            print(f"    {c:40s} {v:16.1f}   ({n} launches)")


if __name__ == "__main__":
    main()
