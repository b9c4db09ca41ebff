#!/usr/bin/env python3
"""tools/kernel_timeline.py RUN.db [N] -- the last N kernel dispatches of a rocprofv3 run in launch order: start offset,
duration, idle gap in front of each (us), short name; and the busy / idle totals of that window (how much of a solve the
card waits for the host)."""
import re, sqlite3, sys
db = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
c = sqlite3.connect(db)
rows = list(c.execute("select start, end, name from kernels order by start"))[-n:]
t0 = rows[0][0]
busy = idle = 0.0
prev_end = rows[0][0]
agg = {}
for s, e, name in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("ddamg::", "")
    short = re.sub(r"<.*", "", short)
    gap = (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {gap:8.1f}  {short}")
    busy += (e - s) / 1e3
    idle += max(gap, 0.0)
    a = agg.setdefault(short, [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] += max(gap, 0.0)
    prev_end = max(prev_end, e)
print(f"# window {(rows[-1][1] - t0) / 1e3:.1f} us: busy {busy:.1f} us, idle {idle:.1f} us")
for k, (cnt, b, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"# {k:40s} {cnt:5d} launches  busy {b:9.1f}  idle in front {g:9.1f}")
