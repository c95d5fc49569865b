import csv, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open((glob.glob(sys.argv[1] + "/*kernel_stats.csv") + glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[0])):
    if pat in r["Name"]: print(f'{r["Name"][:48]:48s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:9.1f} us')
