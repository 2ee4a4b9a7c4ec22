#!/usr/bin/env python3
"""Collect per-kernel HBM traffic (FETCH_SIZE / WRITE_SIZE) of one bench.py run with rocprofv3 and write
profiles/<tag>_pmc_traffic.json.  Run ON THE GPU BOX (via gpurun):

    python tools/collect_pmc.py r01_fft [--mixer fft]

Two separate --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit in one pass), --kernel-trace
only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide coalesced streaming read, so the read side is doubled (field `fetch_bytes_x2`); WRITE_SIZE
is exact for 16-byte-per-lane streaming stores.  bench.py picks the entry of its dominant kernel for `roofline.traffic`.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counter, outdir, mixer):
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-roofline", "--no-every-row", "--no-dp-sequence", "--no-script-leg", "--no-base224",
           "--variants", "none", "--mixer", mixer]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(outdir, "*", "*counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            key = (r["Kernel_Name"], int(r["Grid_Size"]))
            agg[key].append(float(r["Counter_Value"]))
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)  # the counter rows carry the dispatch's stamps
    return {k: (sum(v) / len(v), len(v), sum(dur[k]) / len(dur[k])) for k, v in agg.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    mixer = sys.argv[sys.argv.index("--mixer") + 1] if "--mixer" in sys.argv else "fft"
    scratch = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
    fetch = run_pass("FETCH_SIZE", scratch + "_fetch", mixer)
    write = run_pass("WRITE_SIZE", scratch + "_write", mixer)
    out = []
    for key in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0, 0))[0] + write.get(k, (0, 0, 0))[0])):
        name, grid = key
        f_kib, n, us = fetch.get(key, (0.0, 0, None))
        w_kib, _, _ = write.get(key, (0.0, 0, None))
        short = name.replace("void ", "").replace("(anonymous namespace)::", "")
        short = short.split("(")[0].strip()[:90]
        out.append(dict(kernel=short, grid_size=grid, launches=n, avg_us=None if us is None else round(us, 2), fetch_size_kib=round(f_kib, 1), write_size_kib=round(w_kib, 1),
                        fetch_bytes_x2=int(f_kib * 1024 * 2), write_bytes=int(w_kib * 1024),
                        hbm_bytes_corrected=int(f_kib * 1024 * 2 + w_kib * 1024)))
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_traffic.json")
    json.dump(dict(mixer=mixer, note="per-launch averages; FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE exact", kernels=out[:40]),
              open(path, "w"), indent=1)
    for e in out[:12]:
        print(e)


if __name__ == "__main__":
    main()
