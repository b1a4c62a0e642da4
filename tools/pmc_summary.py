"""Sum rocprofv3 --pmc counter values per kernel.
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> [<second dir> ...]
Prints, per kernel (template arguments kept, parameter lists dropped): dispatches and the total of every counter."""
import csv, glob, os, re, sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").strip()
            tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[name].add((f, row["Dispatch_Id"]))
counters = sorted({c for k in tot for c in tot[k]})
print("kernel".ljust(34), "dispatches", *[c.rjust(18) for c in counters])
for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
    n = len({d for _, d in disp[k]}) if len(sys.argv) == 2 else len(disp[k]) // max(1, len(counters))
    print(k.ljust(34), str(n).rjust(10), *[("%.0f" % tot[k].get(c, 0)).rjust(18) for c in counters])
