"""Per-kernel totals of a rocprofv3 run whose output is a rocpd database (this image's rocprofv3 writes <name>_results.db):
tools/rocpd_kernel_stats.py <results.db> [--csv out.csv]   ->  name, launches, average / total duration, share."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), avg(end - start), sum(end - start), min(end - start), max(end - start) from kernels group by name order by 4 desc").fetchall()
tot = sum(r[3] for r in rows) or 1
lines = ["name,calls,avg_us,total_ms,min_us,max_us,share_pct"]
for r in rows:
    lines.append('"%s",%d,%.2f,%.3f,%.2f,%.2f,%.2f' % (r[0], r[1], r[2] / 1e3, r[3] / 1e6, r[4] / 1e3, r[5] / 1e3, 100.0 * r[3] / tot))
if "--csv" in sys.argv:
    open(sys.argv[sys.argv.index("--csv") + 1], "w").write("\n".join(lines) + "\n")
for r in rows:
    print("%-72s n=%6d avg=%9.1f us total=%8.1f ms %5.1f%%" % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e6, 100.0 * r[3] / tot))
