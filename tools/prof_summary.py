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
