# kernel timeline of the last two solves of the rehearsed 8-GPU problem, self-exchange against plain: busy / idle, per kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rtl; mkdir -p $O
for sx in 1 0; do
rocprofv3 --kernel-trace -d $O/t$sx -o t -- python3 tools/rehearse_profile.py 8 $sx 3 > $O/solve$sx.log 2>$O/err$sx.log
python3 - $O/t$sx/t_results.db $sx <<'PY'
import sqlite3, sys, re
db, sx = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select start, end, name from kernels order by start"))
# the last two solves: from the third-to-last 'vec_from_lex'/'convert' ... simpler: take the last 45 % of the kernels after setup
# find solve boundaries by the fp64 norm of the right-hand side? use dirac_apply_lds_kernel<double launches: 13 per solve (12 its + true residual)
idx = [i for i, r in enumerate(rows) if 'dirac_apply_lds_kernel<double' in r[2]]
per_solve = 13 * (2 if sx == '1' else 1)      # interior + boundary launches on a process grid
start = idx[-2 * per_solve] if len(idx) >= 2 * per_solve else 0
rows = rows[start:]
agg = {}
for s, e, name in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("ddamg::", ""); short = re.sub(r"<.*", "", short)
    a = agg.setdefault(short, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
tot = (rows[-1][1] - rows[0][0]) / 1e3
print(f"# {'self-exchange' if sx == '1' else 'plain'}: window {tot / 2e3:.2f} ms per solve, kernels {sum(v[1] for v in agg.values()) / 2e3:.2f} ms per solve")
for k, (n, b) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"#   {k:38s} {n // 2:5d} launches {b / 2e3:8.2f} ms per solve")
PY
rm -rf $O/t$sx
done
