"""Per-kernel mean of one PMC counter from a rocprofv3 --pmc run (counter_collection.csv): measurement helper.
usage: pmc_summary.py <dir with *_counter_collection.csv> <COUNTER> [name filter]"""
import csv, glob, sys, collections
d, ctr = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == ctr and flt in r["Kernel_Name"]:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v)/len(v):14.1f} mean  {len(v):6d} launches  {sum(v):16.1f} total  {k[:110]}")
