import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
div=float(sys.argv[2])
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%7.3f ms x%6.1f  %s" % (float(r["TotalDurationNs"])/div/1e6, int(r["Calls"])/div, r["Name"][:100]))
print("total", tot/div/1e6)
