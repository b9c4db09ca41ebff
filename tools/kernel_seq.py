#!/usr/bin/env python3
"""tools/kernel_seq.py RUN.db SUBSTR [N] -- durations (us) of the last N dispatches of the kernels whose name contains
SUBSTR, in launch order, and their mean by position modulo PERIOD (env KSEQ_PERIOD, default 4: the four colour launches
of one smoother call)."""
import os, sqlite3, sys
db, sub = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
period = int(os.environ.get("KSEQ_PERIOD", "4"))
c = sqlite3.connect(db)
rows = [r for r in c.execute("select start, end, name from kernels order by start") if sub in r[2]]
rows = rows[-n:]
d = [(e - s) / 1e3 for s, e, _ in rows]
print(" ".join(f"{x:.0f}" for x in d))
for k in range(period):
    v = d[k::period]
    if v:
        print(f"  position {k}: mean {sum(v)/len(v):.1f} us over {len(v)} launches")
