"""Queue / stream of every kernel of the LAST submission in a rocprofv3 rocpd sqlite file.  Usage: python tools/rocpd_queues.py <results.db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("PRAGMA table_info(kernels)").fetchall()]
want = [x for x in ("name", "start", "end", "queue_id", "stream_id", "queue", "stream", "agent_abs_index") if x in cols]
rows = c.execute("select %s from kernels order by start" % ", ".join(want)).fetchall()
ni = want.index("name")
starts = [i for i, r in enumerate(rows) if "k_threshold" in r[ni]]
i0 = starts[-1]
t0 = rows[i0][1]
print("columns:", cols)
for r in rows[i0:]:
    extra = " ".join("%s=%s" % (k, v) for k, v in zip(want[3:], r[3:]))
    print("%-30s start %7.1f end %7.1f us  %s" % (r[ni].split("(")[0][:30], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, extra))
