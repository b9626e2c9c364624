"""Summarise a rocprofv3 results DB: per-kernel calls / total / average (us).  usage: prof_summary.py <db> [n]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = c.execute("select * from top_kernels").fetchall()
tot = sum(r[2] for r in rows)
print('%-90s %7s %12s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', '%'))
for r in rows[:n]:
    print('%-90s %7d %12.1f %10.2f %6.2f' % (r[0][:90], r[1], r[2], r[3], r[4]))
print('total kernel time (us): %.1f over %d kernels' % (tot, sum(r[1] for r in rows)))
