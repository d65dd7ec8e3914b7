"""Kernel dispatches of a rocprofv3 --kernel-trace database (rocpd SQLite) with the gaps between them: where the time between the
kernels of back-to-back vgx_tessellate calls goes.
usage: python profiles/trace_tail.py results.db [N]            last N dispatches
       python profiles/trace_tail.py results.db around NAME K  the dispatches from the K-th to the (K+3)-th launch of kernel NAME"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1]).cursor()
rows = c.execute("select name, start, end from kernels order by start").fetchall()
if len(sys.argv) > 2 and sys.argv[2] == "around":
    hits = [i for i, r in enumerate(rows) if sys.argv[3] in r[0]]
    k = int(sys.argv[4])
    rows = rows[hits[k] - 2:hits[min(k + 3, len(hits) - 1)] + 3]
else:
    rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 16):]
t0 = rows[0][1]
prev = t0
for name, s, e in rows:
    print("%9.1f us  +gap %7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name[:80]))
    prev = e
