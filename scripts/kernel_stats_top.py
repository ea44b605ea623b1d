import csv,sys,glob,collections
for d in sys.argv[1:]:
    f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)
    if not f: print(d,'no stats'); continue
    print(d)
    rows=list(csv.DictReader(open(f[0])))
    for r in rows[:12]:
        print("  %-60s calls=%6s avg_us=%9.1f pct=%5s"%(r['Name'][:60],r['Calls'],float(r['AverageNs'])/1e3,r['Percentage']))
