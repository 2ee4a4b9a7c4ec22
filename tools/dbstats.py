"""Per-kernel totals from a rocprofv3 results .db (sqlite): python tools/dbstats.py path.db [top]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows)
    print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} launches")
    for name, n, t, avg, mn in rows[:top]:
        name = name.replace("void ", "").replace("(anonymous namespace)::", "")
        print(f"{name[:84]:84s} {n:6d} {avg/1e3:8.1f}us min {mn/1e3:7.1f} {t/tot*100:5.1f}%")


if __name__ == "__main__":
    main()
