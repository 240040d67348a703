#!/usr/bin/env python3
"""Print (and optionally write as CSV) per-kernel duration statistics from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME` leaves DIR/NAME_results.db)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
                  "group by name order by 6 desc").fetchall()
lines = ["kernel,calls,avg_us,min_us,max_us,total_us"]
for name, n, avg, mn, mx, tot in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    lines.append("%s,%d,%.2f,%.2f,%.2f,%.2f" % (short, n, avg / 1e3, mn / 1e3, mx / 1e3, tot / 1e3))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
