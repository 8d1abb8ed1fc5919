"""Summarises a rocprofv3 (rocpd sqlite) kernel trace: calls, total/avg/min/max duration per kernel.
Usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, cnt, tot, avg, mn, mx in rows:
        short = n.split("(")[0][:70]
        lines.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (short, cnt, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    # the roofline kernel is launched in two shapes by bench.py (160-frame threshold-only launches for the
    # roofline figure, 256-frame launches inside the pipeline): list them separately so that the average
    # of the roofline launches can be compared with bench.py's roofline.avg_launch_ms
    if "grid_x" in cols:
        rows2 = c.execute("select %s, grid_x, grid_y, grid_z, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
                          "where %s like '%%k_threshold%%' group by %s, grid_x, grid_y, grid_z order by 1, 2" %
                          (name_col, name_col, name_col)).fetchall()
        if rows2:
            lines += ["", "Threshold launches by grid (work-items):", "", "| kernel | grid | calls | avg us | min us | max us |",
                      "|---|---|---|---|---|---|"]
            for n, gx, gy, gz, cnt, avg, mn, mx in rows2:
                lines.append("| %s | %s x %s x %s | %d | %.2f | %.2f | %.2f |" % (n.split("(")[0][:40], gx, gy, gz, cnt, avg / 1e3, mn / 1e3, mx / 1e3))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
