# bootstrap with the coarsest-level solves in lockstep (matrix cores) against one at a time: setup phases at 32^4, kernel statistics
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lockstep; mkdir -p $O
echo "lockstep:"; DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 2 1 32 2 2>&1 | grep -E "bootstrap|Galerkin|smooth|Gram|solve_s" | tail -8
echo "one at a time:"; DDAMG_BOOTSTRAP_NO_LOCKSTEP=1 DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 2 1 32 2 2>&1 | grep -E "bootstrap|Galerkin|smooth|Gram|solve_s" | tail -8
rocprofv3 --kernel-trace --stats -d $O/t -o t -- python3 tools/solve_profile.py 1 1 32 2 > $O/solve.log 2>$O/err.log
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/lockstep/t/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:28]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), r['TotalDurationNs'].rjust(12), r['AverageNs'].rjust(12), r['Percentage'])
PY
cp $O/t/*kernel_stats.csv $O/lockstep_kernel_stats.csv; rm -rf $O/t
