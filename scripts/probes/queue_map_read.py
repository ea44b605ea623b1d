import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
print(" ".join(f"s{r['Stream_Id']}:q{r['Queue_Id']}" for r in rows))
