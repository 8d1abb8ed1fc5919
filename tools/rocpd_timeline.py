"""Timeline of the quad-fit dispatches of the longest step in a rocprofv3 rocpd sqlite file:
start/end of every k_fit_quads launch relative to the first one of its step (shows how the size
classes overlap).  Usage: python tools/rocpd_timeline.py <results.db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
extra = ", grid_x, grid_y, lds_size" if "lds_size" in cols else ", 0, 0, 0"
rows = c.execute("select name, start, end%s from kernels order by start" % extra).fetchall()
# group fit launches that follow a k_scatter (one group per step)
groups, cur = [], None
for r in rows:
    n = r[0]
    if "k_scatter" in n:
        cur = {"t0": r[2], "fits": [], "decode": None}
        groups.append(cur)
    elif cur is not None and ("k_fit_" in n or "k_quad_finish" in n):
        cur["fits"].append(r)
    elif cur is not None and "k_decode_wave" in n and cur["decode"] is None:
        cur["decode"] = r
best = max((g for g in groups if g["fits"] and g["decode"]), key=lambda g: g["decode"][1] - g["t0"])
t0 = best["t0"]
print("step with the longest scatter-end -> decode-start span: %.3f ms" % ((best["decode"][1] - t0) / 1e6))
for r in best["fits"]:
    print("%-24s start %8.3f ms  end %8.3f ms  dur %8.3f ms  grid=(%s,%s) lds=%s" %
          (r[0].split("(")[0][5:], (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[3], r[4], r[5]))
