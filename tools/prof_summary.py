"""Summarise a rocprofv3 rocpd database (top_kernels view) as markdown.  usage: prof_summary.py db title out.md [steps]"""
import sqlite3, sys
db, title, out = sys.argv[1], sys.argv[2], sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else None
rows = sqlite3.connect(db).execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w") as fh:
    fh.write("# %s\n\nrocprofv3 --kernel-trace --stats; durations in microseconds; total GPU kernel time %.1f us" % (title, tot))
    if steps:
        fh.write(" = %.1f us per step over %d steps" % (tot / steps, steps))
    fh.write("\n\n| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
    for n, c, t, a, p in rows:
        n = n if len(n) < 120 else n[:117] + "..."
        fh.write("| `%s` | %d | %.1f | %.1f | %.2f |\n" % (n.replace("|", "/"), c, t, a, p))
for n, c, t, a, p in rows[:22]:
    print("%-90s %5d %10.1f %8.1f %6.2f" % (n[:90], c, t, a, p))

# ---- optional detail sections (when the database has the `kernels` view): per-(kernel, grid) averages of the GEMM kernels and
# ---- one timed training step in launch order (the launches between two consecutive optimizer kernels in the middle of the run)
try:
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else "grid_size_x"
    with open(out, "a") as fh:
        fh.write("\n## GEMM launches by grid (threads in x)\n\n| kernel | grid x | calls | avg us |\n|---|---|---|---|\n")
        q = "select name, %s, count(*), avg(end - start) / 1000.0 from kernels where name like '%%gemm%%' group by name, %s order by 1, 2" % (gx, gx)
        for n, g, c, a in con.execute(q):
            fh.write("| `%s` | %d | %d | %.1f |\n" % (n[:60].replace("|", "/"), g, c, a))
        adam = [r[0] for r in con.execute("select start from kernels where name like '%adam%' order by start")]
        if len(adam) >= 4:
            j = len(adam) // 2
            fh.write("\n## one training step in launch order (between optimizer launches %d and %d)\n\n| t us | dur us | grid x | kernel |\n|---|---|---|---|\n" % (j, j + 1))
            rows_ = con.execute("select name, %s, start, end from kernels where start > ? and start <= ? order by start" % gx, (adam[j], adam[j + 1])).fetchall()
            for n, g, s_, e_ in rows_:
                fh.write("| %.1f | %.1f | %d | `%s` |\n" % ((s_ - rows_[0][2]) / 1000.0, (e_ - s_) / 1000.0, g, n[:70].replace("|", "/")))
except Exception as exc:      # the summary above stands on its own
    print("detail sections skipped:", exc)
