#!/usr/bin/env python3
"""tools/pmc_summary.py DIR [DIR ...] -- per-kernel mean of every counter found in rocprofv3
counter_collection.csv files under the given directories (one --pmc pass per directory), as JSON."""
import csv, glob, json, os, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if name.startswith("void "):
                name = name[5:]
            name = name.split("(")[0]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    out[k] = {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()}
print(json.dumps(out, indent=1))
