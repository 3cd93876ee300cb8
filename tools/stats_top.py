import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    print("%9.1f us x%4s %5.1f%%  %s" % (float(r["AverageNs"]) / 1000, r["Calls"], float(r["Percentage"]), r["Name"][:100]))
