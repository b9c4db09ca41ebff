#!/usr/bin/env python3
"""tools/rocpd_export.py -- turn rocprofv3's rocpd sqlite output into the small text summaries kept under profiles/.

  rocpd_export.py stats  RUN.db  OUT.csv     per-kernel calls / total / average duration (the --stats table)
  rocpd_export.py pmc    RUN.db [RUN.db ...] per-kernel mean of every collected counter, JSON on stdout
"""
import csv, json, sqlite3, sys
from collections import defaultdict


def short(name):
    if name.startswith("void "):
        name = name[5:]
    return name.replace("(anonymous namespace)::", "").split("(")[0]


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.0f}", f"{r[3]:.1f}", f"{r[4]:.3f}"])


def pmc(dbs):
    acc = defaultdict(lambda: defaultdict(list))
    for db in dbs:
        c = sqlite3.connect(db)
        for name, counter, value in c.execute("select kernel_name, counter_name, value from counters_collection"):
            acc[short(name)][counter].append(float(value))
    out = {k: {cn: {"mean": sum(v) / len(v), "launches": len(v)} for cn, v in cs.items()} for k, cs in sorted(acc.items())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:])
