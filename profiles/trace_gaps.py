"""Per-kernel start offsets of ONE vgx_tessellate call from a rocprofv3 --kernel-trace database (rocpd SQLite):
shows where a small call's time goes (kernel durations vs the gaps between dependent kernels)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1]).cursor()
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# take the last complete call: from the last 'OpCmdPrefix' reduce/single scan to the following k_publish
starts = [i for i, r in enumerate(rows) if "OpCmdPrefix" in r[0] and ("k_scan_reduce" in r[0] or "k_scan_single" in r[0])]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
i1 = next(i for i in range(i0, len(rows)) if "k_publish" in rows[i][0])
t0 = rows[i0][1]
prev_end = t0
for name, s, e in rows[i0:i1 + 1]:
    print("%8.1f us  +gap %6.1f  dur %6.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name[:90]))
    prev_end = e
print("call span %.1f us" % ((rows[i1][2] - t0) / 1e3))
