import csv,sys,glob
for f in sys.argv[1:]:
    for p in glob.glob(f+"/**/*kernel_stats.csv", recursive=True):
        rows=list(csv.DictReader(open(p)))
        print(f)
        for r in rows[:6]:
            print("   %-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
