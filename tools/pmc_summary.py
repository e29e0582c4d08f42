"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per dispatch."""
import csv, sys, collections
path, want = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
    if want and not any(w in name for w in want):
        continue
    acc[name.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        print("%-62s %-12s n=%6d mean=%.6g sum=%.6g" % (k, c, len(v), sum(v) / len(v), sum(v)))
