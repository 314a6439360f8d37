"""Dispatch-ordered kernel list from a rocprofv3 --kernel-trace sqlite database: python tools/ktrace_list.py <db> [name filter]
Prints start offset, duration, grid and workgroup size of every dispatch (for launches that one call splits into several kernels)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
gx = "d.grid_size_x" if "grid_size_x" in cols else ("d.grid_x" if "grid_x" in cols else "0")
wx = "d.workgroup_size_x" if "workgroup_size_x" in cols else ("d.workgroup_x" if "workgroup_x" in cols else "0")
rows = c.execute(f"select s.kernel_name, d.start, d.end, {gx}, {wx} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
t0 = rows[0][1] if rows else 0
for n, s, e, g, w in rows:
    if flt and flt not in n:
        continue
    n = re.sub(r"\.kd$", "", n)
    m = re.search(r"gemm_nt_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)", n)
    tag = ("nt %sx%s epi%s bf16=%s persist=%s direct=%s" % (m.group(1), m.group(2), m.group(8), m.group(9), m.group(10), m.group(11))) if m else n[:60]
    print(f"{(s - t0) / 1e3:12.1f} us  {(e - s) / 1e3:8.1f} us  grid {g:>8} wg {w:>4}  {tag}")
