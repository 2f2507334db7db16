"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into per-kernel stats.
usage: python profiles/summarize_rocpd.py path/to/results.db [n_steps]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                 "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot / 1e3:.2f} ms" + (f" over {steps} steps = {tot / 1e3 / steps:.3f} ms/step" if steps else ""))
print(f"{'kernel':92s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for r in rows:
    print(f"{r[0][:92]:92s} {r[1]:6d} {r[2] / 1e3:10.3f} {r[3]:10.2f} {r[4]:9.2f} {r[5]:9.2f} {100 * r[2] / tot:6.1f}")
