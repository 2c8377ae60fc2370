"""Summarise a rocprofv3 counter_collection.csv: per kernel name, mean of each counter over its dispatches."""
import collections, csv, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:90]
    rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, cs in rows.items():
    n = len(next(iter(cs.values())))
    print(f"{k}  dispatches={n}  ms(mean)={sum(dur[k]) / len(dur[k]):.3f}")
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {sum(v) / len(v):16.0f}")
