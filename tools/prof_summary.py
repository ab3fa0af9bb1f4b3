#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel name, calls, total / average / min / max duration in us)."""
import glob, sqlite3, sys
path = sys.argv[1]
dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
for db in dbs:
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# {db}: {len(rows)} kernels, total {tot/1e3:.3f} ms")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  name")
    for name, n, t, a, mn, mx in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        print(f"{n:6d} {t:12.1f} {a:10.2f} {mn:10.2f} {mx:10.2f} {100*t/tot:6.2f}  {short}")
