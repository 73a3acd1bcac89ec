"""Print per-kernel PMC counter values from a rocprofv3 rocpd sqlite db (average over dispatches of kernels matching a pattern)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor(); pat = sys.argv[2] if len(sys.argv) > 2 else 'p5_'
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
ev = [t for t in tabs if 'pmc_event' in t][0]; info = [t for t in tabs if 'info_pmc' in t][0]
disp = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'info_kernel_symbol' in t][0]
rows = cur.execute(f"select s.kernel_name, i.name, avg(e.value), count(*) from {ev} e join {info} i on e.pmc_id=i.id join {disp} d on e.event_id=d.event_id "
                   f"join {sym} s on d.kernel_id=s.id where s.kernel_name like '%{pat}%' group by s.kernel_name, i.name").fetchall()
for k, n, v, c in rows:
    print(f"{n:32s} {v:16.1f}  (n={c})  {k[:70]}")
