"""Start / end of every kernel of the LAST submission in a rocprofv3 rocpd sqlite file, relative to its first kernel
(how the launches of a one-frame submission follow each other).  Usage: python tools/rocpd_last_step.py <results.db> [back]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = c.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "k_threshold" in r[0] and "leftover" not in r[0]]
i0 = starts[-1 - back]
i1 = starts[-back] if back else len(rows)
t0 = rows[i0][1]
prev_end = t0
for r in rows[i0:i1]:
    print("%-34s start %8.1f us  end %8.1f us  dur %7.1f us  gap-after-prev-end %6.1f" %
          (r[0].split("(")[0][:34], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3))
    prev_end = max(prev_end, r[2])
