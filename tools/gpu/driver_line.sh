# the driver's invocation of bench.py, timed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/driver_line
t0=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_line/line.json 2> gpurun_out/driver_line/err.log; echo "rc=$? wall $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s"; python3 -c "
import json; d=json.loads(open('gpurun_out/driver_line/line.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','vs_baseline','dtype','data')})
print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline']['kind'], d['cpu_baseline']['cores'])
print({k:v for k,v in d['solve'].items() if k not in ('workload','coarse_operator')}); print(d['solve']['coarse_operator']['solve_path'])
print({k:v for k,v in d['strong_scaling'].items() if k!='workload'})"
