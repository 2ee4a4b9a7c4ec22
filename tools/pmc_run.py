#!/usr/bin/env python3
"""Per-kernel hardware counters for any python script (development tool; run ON THE GPU BOX via gpurun).

    python tools/pmc_run.py "CTR_A CTR_B,CTR_C CTR_D" [--filter substr] -- tools/gemm_check.py 33280 768 512

Comma-separated groups are separate rocprofv3 --pmc passes (--kernel-trace only, as MI355X_MICROARCH.md prescribes).
Prints the per-launch average of every counter for each kernel whose name contains the filter.

Every group is checked against the per-block counter slots of gfx950 BEFORE anything runs and split when it does not fit
(SLOTS below: SQ 8, TCC 4, GRBM 2 from MI355X_MICROARCH.md "rocprofv3 PMC slots"; TA / TD 2 and TCP 4 are the
TA_PERFCOUNTER0-1 / TD_PERFCOUNTER0-1 / TCP_PERFCOUNTER0-3 select registers of the gfx9 family).  Round 1 lost a pass to
this: "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" asks for three TA events on two TA
slots; rocprofv3 did not reject it, the profiled program never finished and the pass was killed at this tool's limit,
silently.  A pass that times out or fails now prints what it captured and makes the tool exit non-zero.
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

SLOTS = {"SQ": 8, "TCC": 4, "GRBM": 2, "TA": 2, "TD": 2, "TCP": 4, "CPC": 2, "CPF": 2, "SPI": 4}
COST = {"FETCH_SIZE": ("TCC", 3), "WRITE_SIZE": ("TCC", 2)}  # derived metrics that take several raw counters


def block_cost(counter):
    if counter in COST:
        return COST[counter]
    return counter.split("_")[0], 1


def fit_groups(groups):
    """split every requested pass so that no hardware block is asked for more events than it has counter slots"""
    out = []
    for g in groups:
        cur, used = [], {}
        for c in g:
            blk, n = block_cost(c)
            cap = SLOTS.get(blk, 2)  # unknown block: assume the smallest bank
            if n > cap:
                raise SystemExit(f"{c} needs {n} {blk} slots, the block has {cap}")
            if used.get(blk, 0) + n > cap:
                out.append(cur)
                cur, used = [], {}
            cur.append(c)
            used[blk] = used.get(blk, 0) + n
        if cur:
            out.append(cur)
    return out


def main():
    args = sys.argv[1:]
    sep = args.index("--")
    asked = [g.split() for g in args[0].split(",")]
    groups = fit_groups(asked)
    if groups != asked:
        print(f"regrouped to fit the counter slots: {groups}", flush=True)
    failed = 0
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
            stdout, _ = proc.communicate()
            print(f"pass {g} TIMED OUT after 100 s; captured output:\n{(stdout or '')[-3000:]}", flush=True)
            failed += 1
            break  # no further GPU step after a killed one
        if proc.returncode != 0:
            print(f"pass {g} failed (rc {proc.returncode}):\n{stdout[-3000:]}", flush=True)
            failed += 1
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
    if failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
