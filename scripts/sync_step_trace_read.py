import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into steps by gaps > 5 ms
steps, cur, last = [], [], None
for r in rows:
    s = int(r["Start_Timestamp"])
    if last is not None and s - last > 5_000_000:
        steps.append(cur); cur = []
    cur.append(r); last = int(r["End_Timestamp"])
steps.append(cur)
for st in steps[-2:]:
    t0 = int(st[0]["Start_Timestamp"])
    print("step with", len(st), "kernels, span", (int(st[-1]["End_Timestamp"]) - t0) / 1e6, "ms")
    keyname = "(Stream_Id, Queue_Id)"
    by = collections.defaultdict(list)
    for r in st:
        by[(r["Stream_Id"], r["Queue_Id"])].append(r)
    for k, v in by.items():
        print("  ", keyname, k, "kernels", len(v), "first start %.3f ms" % ((int(v[0]["Start_Timestamp"]) - t0) / 1e6), "last end %.3f ms" % ((int(v[-1]["End_Timestamp"]) - t0) / 1e6),
              "busy %.3f ms" % (sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v) / 1e6))
    tiles = [r for r in st if "k3_tile_all" in r["Kernel_Name"]]
    print("   first 12 tile kernels start at (ms):", ["%.3f" % ((int(r["Start_Timestamp"]) - t0) / 1e6) for r in tiles[:12]])
