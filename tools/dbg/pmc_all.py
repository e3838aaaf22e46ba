import csv, collections, glob, os, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:22]
names = sorted({n for _, c in rows for n in c if n != "SQ_WAVE_CYCLES"})
print("%-62s %14s " % ("kernel", "WAVE_CYCLES") + " ".join("%20s" % n.replace("SQ_", "") for n in names))
for k, c in rows:
    b = c["SQ_WAVE_CYCLES"]
    print("%-62s %14.3e " % (k, b) + " ".join("%20.3f" % (c.get(n, 0) / b) for n in names))
