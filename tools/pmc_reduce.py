"""Reduce a rocprofv3 --pmc output directory to kernel,counter,dispatches,mean_per_dispatch."""
import collections
import csv
import glob
import sys

f = glob.glob(f"{sys.argv[1]}/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,counter,dispatches,mean_per_dispatch")
for (k, c), v in sorted(agg.items()):
    print(f'"{k}",{c},{len(v)},{sum(v) / len(v):.3f}')
