import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    if "bnn::" not in k: continue
    print(k, "dispatches", len(list(v.values())[0]), "dur_us(profiled) %.1f" % (sum(dur[k]) / len(dur[k]) / 1e3))
    for c, x in sorted(v.items()):
        print("    %-28s %14.1f" % (c, sum(x) / len(x)))
