# bench.py with the rehearsal leg (64^4 strong-scaling N = 1 point + the per-GPU problem of the 8-GPU decomposition)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 200 --warmup 50 --no-solve --no-cpu-baseline > gpurun_out/rehearse.json 2> gpurun_out/rehearse.err; tail -3 gpurun_out/rehearse.err
python3 -c "
import json; d=json.loads(open('gpurun_out/rehearse.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('strong_scaling'), indent=1)[:1500]); print(json.dumps(d.get('rehearsal'), indent=1))"
