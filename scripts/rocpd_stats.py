"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`) as the
per-kernel stats table (name, calls, total/avg/min/max ns, %) and write it as CSV + markdown."""
import sqlite3
import sys

db, out_prefix = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("""select name, count(*), sum(duration), avg(duration), min(duration), max(duration)
                    from kernels group by name order by sum(duration) desc""").fetchall()
total = sum(r[2] for r in rows) or 1
with open(out_prefix + "_kernel_stats.csv", "w") as f:
    f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for n, calls, tot, avg, mn, mx in rows:
        f.write(f'"{n}",{calls},{tot},{avg:.1f},{100.0 * tot / total:.2f},{mn},{mx}\n')
with open(out_prefix + "_kernel_stats.md", "w") as f:
    f.write("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|\n")
    for n, calls, tot, avg, mn, mx in rows[:25]:
        f.write(f"| `{n[:110]}` | {calls} | {avg / 1e6:.4f} | {tot / 1e6:.3f} | {100.0 * tot / total:.2f} |\n")
print(open(out_prefix + "_kernel_stats.md").read())
