# after the workload-dependent default of the reserved CUs: multi-process / self-exchange tests, then the rehearsal and the apply
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests/test_multi_process.py tests/test_gpu_self_exchange.py tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -3 &&
python3 bench.py --steps 500 --warmup 100 --no-solve --no-strong --no-cpu-baseline --self-exchange=-1,-1,-1,1 2>&1 | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("apply us (default)", round(d["ms_per_step"]*1000,1))' &&
python3 tools/rehearse_profile.py 8 1 3 | tail -1 | cut -c1-150
