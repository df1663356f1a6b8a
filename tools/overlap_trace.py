"""How much of the grouped weight-gradient launch really runs beside other kernels: from a rocprofv3 --kernel-trace database (rocpd)
of tools/step_one.py with STEP_OPTS=overlap_wgrad=1, print one training step in START order with start / end times and, for every
ltrx_gemm_tn256_kernel launch, the kernels whose execution intervals intersect it and by how much.
    usage: overlap_trace.py db [out.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else "grid_size_x"
extra = [c for c in ("queue_id", "stream_id") if c in cols]
adam = [r[0] for r in db.execute("select start from kernels where name like '%adam%' order by start")]
j = len(adam) - 2                       # the last complete step
rows = db.execute("select name, %s, start, end %s from kernels where start > ? and start <= ? order by start"
                  % (gx, "".join(", " + c for c in extra)), (adam[j], adam[j + 1])).fetchall()
t0 = rows[0][2]
span = (max(r[3] for r in rows) - t0) / 1000.0
busy = sum(r[3] - r[2] for r in rows) / 1000.0
print("## one training step in start order (last complete step of the trace)\n", file=out)
print("wall %.1f us from the first kernel's start to the last kernel's end; sum of kernel durations %.1f us (ratio %.3f: > 1 means kernels ran side by side)\n"
      % (span, busy, busy / span), file=out)
print("| start us | end us | dur us | grid x | %s kernel |\n|---|---|---|---|%s---|" % ("".join(c + " | " for c in extra), "---|" * len(extra)), file=out)
for r in rows:
    print("| %.1f | %.1f | %.1f | %d | %s`%s` |" % ((r[2] - t0) / 1000.0, (r[3] - t0) / 1000.0, (r[3] - r[2]) / 1000.0, r[1],
                                                 "".join("%s | " % (v,) for v in r[4:]), r[0][:60].replace("|", "/")), file=out)
print("\n## what ran while a grouped weight-gradient launch (ltrx_gemm_tn256_kernel) was executing\n", file=out)
for r in rows:
    if "gemm_tn256" not in r[0]:
        continue
    s, e = r[2], r[3]
    tot = 0.0
    print("* tn256 launch %.1f .. %.1f us (%.1f us):" % ((s - t0) / 1000.0, (e - t0) / 1000.0, (e - s) / 1000.0), file=out)
    for q in rows:
        if q is r:
            continue
        ov = min(e, q[3]) - max(s, q[2])
        if ov > 0:
            tot += ov
            print("    - `%s` (grid %d) overlaps %.1f of its %.1f us" % (q[0][:50], q[1], ov / 1000.0, (q[3] - q[2]) / 1000.0), file=out)
    print("    total overlapped kernel time %.1f us" % (tot / 1000.0), file=out)
