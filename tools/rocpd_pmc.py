"""Prints per-dispatch PMC counter values for kernels matching a substring (rocprofv3 rocpd sqlite)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
for t in ("pmc_events", "counters_collection"):
    try:
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
        print(t, cols)
    except Exception as e:
        print(t, e)
try:
    rows = c.execute("select * from counters_collection where kernel_name like ? limit 40", ("%" + pat + "%",)).fetchall()
    for r in rows: print(r)
except Exception as e:
    print("query failed", e)
