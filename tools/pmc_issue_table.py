"""per-kernel issue utilisation from a tools/pmc_run.py log: VALU per wave, share of a wave's lifetime spent issuing / parked / stalled"""
import sys
d, cur = {}, None
for l in open(sys.argv[1]).read().split("\n"):
    if l and not l.startswith(" ") and not l.startswith("pass") and not l.startswith("regrouped") and not l.startswith("/"):
        cur = l.strip()
        d[cur] = {}
    elif l.startswith("    ") and cur:
        p = l.split()
        try:
            d[cur][p[0]] = float(p[1])
        except (ValueError, IndexError):
            pass
print(f"{'kernel':62s} {'waves':>7s} {'VALU/wave':>9s} {'LDS/wave':>8s} {'active%':>8s} {'parked%':>8s} {'stalled%':>8s} {'cycles/wave':>11s}")
for k, c in d.items():
    if c.get("SQ_WAVES", 0) < 512 or "SQ_WAVE_CYCLES" not in c:
        continue
    w, wc = c["SQ_WAVES"], c["SQ_WAVE_CYCLES"]
    print(f"{k[:62]:62s} {w:7.0f} {c.get('SQ_INSTS_VALU', 0) / w:9.0f} {c.get('SQ_INSTS_LDS', 0) / w:8.0f} {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:8.1f} "
          f"{100 * c.get('SQ_WAIT_ANY', 0) / wc:8.1f} {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:8.1f} {4 * wc / w:11.0f}")
