"""Per-kernel mean of every collected PMC counter (rocprofv3 rocpd sqlite). Usage: db substring"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
rows = c.execute("select kernel_name, counter_name, value, duration, grid_size, lds_block_size from counters_collection where kernel_name like ?", ("%" + pat + "%",)).fetchall()
acc = collections.defaultdict(list)
for k, cn, v, d, g, l in rows:
    acc[(k.split("(")[0][:32], l, cn)].append(v)
for key in sorted(acc):
    v = acc[key]
    print("%-34s lds=%-7s %-24s n=%-3d mean=%.4g" % (key[0], key[1], key[2], len(v), sum(v) / len(v)))
