"""per-(kernel, grid) launch statistics from a rocprofv3 rocpd database.  usage: prof_by_grid.py db [name-filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
if "kernels" not in names:
    print("no `kernels` view; objects:", names)
    sys.exit(1)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
if gx is None or "start" not in cols:
    print("columns:", cols)
    sys.exit(1)
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "1")
q = "select name, %s, %s, count(*), avg(end - start) / 1000.0, sum(end - start) / 1000.0 from kernels where name like ? group by name, %s order by 6 desc" % (gx, wx, gx)
for n, g, w, c, a, t in db.execute(q, ("%" + flt + "%",)).fetchall()[:40]:
    print("%-60s grid %8s wg %5s  calls %4d  avg %8.1f us  total %10.1f us" % (n[:60], g, w, c, a, t))
