"""Summarise a rocprofv3 --pmc output directory: per kernel-name mean of each counter (CSV counter_collection files)."""
import csv, glob, json, re, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"([A-Za-z_0-9]+_kernel(?:<[^>(]*>)?)", row["Kernel_Name"])
        name = m.group(1) if m else row["Kernel_Name"].split("(")[0][:60]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": max(len(v) for v in d.values())} for k, d in acc.items()}
print(json.dumps(out, indent=1))
