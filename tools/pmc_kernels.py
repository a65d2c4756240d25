"""Mean PMC counter values per (kernel, grid size) from rocprofv3 counter_collection.csv files.
    python tools/pmc_kernels.py <substring> <counter_collection.csv>..."""
import collections
import csv
import sys

sub = sys.argv[1]
agg = collections.defaultdict(list)
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        if sub in name:
            agg[(name, int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k[0]:22s} grid {k[1]:9d} {k[2]:28s} {sum(v) / len(v):14.4g}  (n={len(v)})")
