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
    # GPU idle between consecutive dispatches (device timeline): gaps above 1 ms are host phases (setup, the bench's barriers), the
    # rest is what launch latency / kernel drain costs inside the steps
    try:
        tl = con.execute("select start, end from kernels order by start").fetchall()
        gaps = [max(0, tl[i + 1][0] - max(t[1] for t in tl[max(0, i - 3): i + 1])) for i in range(len(tl) - 1)]
        small = [g for g in gaps if g < 1_000_000]
        busy = sum(e - s_ for s_, e in tl)
        f.write(f"\ninter-kernel idle (gaps < 1 ms): {sum(small)/1e6:.2f} ms over {len(small)} boundaries = {100*sum(small)/max(1,busy):.2f} % of the "
                f"kernel time; median {sorted(small)[len(small)//2]/1e3:.2f} us, {sum(1 for g in small if g > 10_000)} gaps above 10 us "
                f"({sum(g for g in small if g > 10_000)/1e6:.2f} ms)\n")
    except Exception as ex:  # noqa: BLE001
        f.write(f"\n(no timeline columns in this database: {ex})\n")
print(open(out).read()[:6000])
