import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(dict)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(d)
sel = [ids[i] for i in map(int, sys.argv[3].split(","))] if len(sys.argv) > 3 else ids[:3]
for i in sel:
    print(i, {k: int(v) for k, v in d[i].items()})
