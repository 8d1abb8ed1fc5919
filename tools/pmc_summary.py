"""Per-kernel mean PMC counter values of a rocprofv3 --pmc run (rocpd sqlite).  Prints one markdown table.
Usage: python tools/pmc_summary.py <dir-or-db> [substring ...]"""
import collections, glob, os, sqlite3, sys
path = sys.argv[1]
pats = sys.argv[2:] or ["k_fit_quads", "k_cc_", "k_points", "k_scatter", "k_cluster_select", "k_decode", "k_threshold"]
dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
acc = collections.defaultdict(list)
dur = collections.defaultdict(list)
for db in dbs:
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select kernel_name, counter_name, value, duration, grid_size, lds_block_size from counters_collection").fetchall()
    except Exception as e:
        print("skip", db, e)
        continue
    for k, cn, v, d, g, l in rows:
        if not any(p in k for p in pats):
            continue
        name = k.split("(")[0].replace("void ", "")[:28]
        if "fit_quads" in name:
            name += " lds=%s" % l
        acc[(name, cn)].append(v)
        dur[name].append(d)
names = sorted({k[0] for k in acc})
ctrs = sorted({k[1] for k in acc})
print("| kernel | launches | avg ms (under PMC) | " + " | ".join(ctrs) + " |")
print("|---|---|---|" + "---|" * len(ctrs))
for n in names:
    d = dur[n]
    nl = max(len(acc[(n, c)]) for c in ctrs if (n, c) in acc)
    print("| %s | %d | %.3f | " % (n, nl, (sum(d) / len(d)) / 1e6) + " | ".join(("%.4g" % (sum(acc[(n, c)]) / len(acc[(n, c)]))) if (n, c) in acc else "-" for c in ctrs) + " |")
