#!/usr/bin/env python3
"""tools/gpu_busy.py RUN.db SECONDS -- share of the last SECONDS of a rocprofv3 --kernel-trace run during which a kernel
was executing (union of the dispatch intervals), and the largest contributors to the idle time by the kernel that follows
the gap."""
import sqlite3, sys
from collections import defaultdict
db, secs = sys.argv[1], float(sys.argv[2])
c = sqlite3.connect(db)
tab = [r[0] for r in c.execute("select name from sqlite_master where type='view' or type='table'") if r[0] == "kernels"]
rows = c.execute("select start, end, name from kernels order by start").fetchall()
t_end = max(r[1] for r in rows); t0 = t_end - int(secs * 1e9)
rows = [r for r in rows if r[0] >= t0]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
gaps = defaultdict(lambda: [0, 0])
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        g = gaps[n.split("(")[0][-60:]]; g[0] += s - cur_e; g[1] += 1
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f"window {span/1e6:.1f} ms, kernels {len(rows)}, busy {busy/span:.3f}")
for n, (t, k) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  idle before {n}: {t/1e6:.2f} ms in {k} gaps ({t/k/1e3:.1f} us each)")
