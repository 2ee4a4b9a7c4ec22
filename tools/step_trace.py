"""Ordered kernel list of ONE training step (names, durations, gaps) out of a rocprofv3 --kernel-trace csv.
   rocprofv3 --kernel-trace [--memory-copy-trace] -d out -o t --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --variants none --no-roofline ...
   python3 tools/step_trace.py out/.../t_kernel_trace.csv [step_index_from_end] [out/.../t_memory_copy_trace.csv]
With the third argument the memory copies (hipMemcpyAsync nodes: SDMA / blit, not in the kernel trace) are merged into the list."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["_name"] = r["Kernel_Name"]
    if len(sys.argv) > 3:
        for r in csv.DictReader(open(sys.argv[3])):
            r["_name"] = f"[memcpy {r.get('Direction', '?')} {r.get('Bytes', r.get('Size', '?'))} B]"
            rows.append(r)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["_name"] for r in rows]
    # a step starts at the patchify kernel
    starts = [i for i, n in enumerate(names) if "patchify" in n]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    a, b = starts[-k - 1], starts[-k]
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["_name"]
        n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
        print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {n[:110]}")
        prev_end = max(prev_end, e)
        busy += e - s
    span = int(rows[b]["Start_Timestamp"]) - t0
    print(f"step span {span / 1e3:.1f} us, kernels {b - a}, busy {busy / 1e3:.1f} us")


if __name__ == "__main__":
    main()
