"""the last N kernel dispatches of a rocprofv3 rocpd database in launch order.  usage: prof_sequence.py db N [name-filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else "grid_size_x"
rows = db.execute("select name, %s, start, end from kernels where name like ? order by start desc limit ?" % gx, ("%" + flt + "%", n)).fetchall()[::-1]
t0 = rows[0][2]
for name, g, s, e in rows:
    print("%9.1f us  +%8.1f us  grid %8d  %s" % ((s - t0) / 1000.0, (e - s) / 1000.0, g, name[:70]))
