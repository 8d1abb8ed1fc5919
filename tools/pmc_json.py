"""Per-kernel FETCH_SIZE / WRITE_SIZE totals of one pipeline pass under rocprofv3 --pmc (rocpd sqlite), as JSON for
bench.py's stage_roofline (profiles/r03_pipeline_pmc.json).  Usage: python tools/pmc_json.py <dir-with-dbs> <frames> <out.json>
Values are the counters as reported (KB per pass over <frames> frames, summed over the launches of a kernel in ONE
submission -- the profiled one, the last); bench.py applies the gfx950 correction (FETCH_SIZE x 2, MI355X_MICROARCH.md)."""
import collections, glob, json, os, sqlite3, sys
path, frames, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]   # frames per submission; pipeline_once.py <B> 1 <d> makes 2 submissions
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for db in sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection order by dispatch_id").fetchall()
    except Exception as e:
        print("skip", db, e)
        continue
    for k, cn, v, did in rows:
        if cn not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        name = k.split("(")[0].replace("void ", "")
        if "<" in name:
            name = name.split("<")[0]
        acc[name][cn].append(v)
rec = {"frames": frames, "submissions": 2, "unit": "KB as reported by rocprofv3 (FETCH_SIZE uncorrected)", "kernels": {}}
for name, d in sorted(acc.items()):
    if not name.startswith("k_"):
        continue
    # pipeline_once runs reps + 1 submissions: a kernel with L launches per submission appears L * (reps + 1) times
    rec["kernels"][name] = {cn: {"launches_seen": len(v), "sum_KB": float(sum(v))} for cn, v in d.items()}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec)[:600])
