# kernel statistics of the rehearsed 8-GPU problem (setup + 2 solves), self-exchange against plain
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/dsetup; mkdir -p $O
for sx in 1 0; do
  rocprofv3 --kernel-trace --stats -d $O/t$sx -o t -- python3 tools/rehearse_profile.py 8 $sx 1 > $O/log$sx.txt 2>$O/err$sx.txt
  python3 tools/rocpd_export.py stats $O/t$sx/t_results.db $O/stats$sx.csv; rm -rf $O/t$sx
done
python3 - <<'PY'
import csv
for sx in (1,0):
    rows=list(csv.DictReader(open(f'gpurun_out/dsetup/stats{sx}.csv')))
    print('self-exchange' if sx else 'plain', 'total ms', sum(float(r['TotalDurationUs']) for r in rows)/1e3)
    for r in rows[:22]:
        n=r['Name'].replace('void ddamg::','').replace('ddamg::','')
        print('  ', n[:60].ljust(60), r['Calls'].rjust(6), f"{float(r['TotalDurationUs'])/1e3:8.1f} ms", r['AverageUs'].rjust(9))
PY
