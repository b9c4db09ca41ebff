# the strong-scaling curve of the 64^4 solve as rehearsed on one GPU: N = 2, 4, 8 (bench.py --rehearse N)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 2 4 8; do
python3 bench.py --steps 50 --warmup 10 --no-solve --no-cpu-baseline --rehearse $n 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['rehearsal']; s=d['strong_scaling']
print('N', h.get('n_gpus_rehearsed'), 'local', h.get('local_lattice'), 'n1', round(s['seconds_per_solve'],4), 'per gpu', round(h['seconds_per_solve_per_gpu'],4), 'plain', round(h['same_lattice_without_the_machinery']['seconds_per_solve'],4), 'predicted', round(h['predicted_seconds_per_solve_per_gpu'],4), 'speedup', round(h['predicted_speedup_vs_n1'],2), 'its', h['iterations'], 'setup', round(h['setup_seconds'],2), 'err', h.get('error'))"
done
