"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table (markdown + csv)."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                   "group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary ({db.split('/')[-1]}); durations in microseconds\n\n")
    f.write("| kernel | calls | total_us | avg_us | min_us | max_us | pct |\n|---|---|---|---|---|---|---|\n")
    for n, c, s, a, mn, mx in rows:
        f.write(f"| `{n[:110]}` | {c} | {s/1e3:.1f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/tot:.2f} |\n")
    f.write(f"\ntotal kernel time {tot/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches\n")
print(open(out).read()[:6000])
