import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
d = collections.defaultdict(list)
for r in rows:
    if pat in r["Kernel_Name"]:
        key = (r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v), "avg us %.1f" % (sum(v) / len(v)))
