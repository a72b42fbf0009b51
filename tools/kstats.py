import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0].replace("void ","")
    if n.startswith("k_"): print("  %-24s calls %3s avg %8.1f us"%(n[:24],r["Calls"],float(r["AverageNs"])/1e3))
