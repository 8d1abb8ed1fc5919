"""Lists individual dispatches of kernels matching a substring from a rocprofv3 rocpd sqlite file."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]; last = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end, grid_x, grid_y, lds_size from kernels where name like ? order by start", ("%" + pat + "%",)).fetchall() \
    if "lds_size" in cols else c.execute("select name, start, end, 0,0,0 from kernels where name like ? order by start", ("%" + pat + "%",)).fetchall()
for r in rows[-last:]:
    print("%-40s %10.1f us grid=(%s,%s) lds=%s" % (r[0].split("(")[0][:40], (r[2] - r[1]) / 1e3, r[3], r[4], r[5]))
print(cols)
