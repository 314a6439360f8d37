"""Per-kernel table from a rocprofv3 --kernel-trace results database (sqlite): python tools/kstats.py <db> <steps incl. warm-up> [rows]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2])
rows_max = int(sys.argv[3]) if len(sys.argv) > 3 else 30
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# total kernel time {tot / 1e6:.1f} ms = {tot / 1e6 / steps:.2f} ms/step over {steps} steps; {sum(r[1] for r in rows) / steps:.0f} launches/step")
print(f"{'kernel':72s} {'calls':>6s} {'ms/step':>9s} {'avg us':>9s} {'share':>6s}")
for n, cnt, t in rows[:rows_max]:
    n = re.sub(r"\.kd$", "", n)
    print(f"{n[:72]:72s} {cnt:6d} {t / 1e6 / steps:9.3f} {t / cnt / 1e3:9.1f} {100.0 * t / tot:5.1f}%")
