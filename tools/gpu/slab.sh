cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_multigrid.py -x -q -m gpu -k "slab or 256_site" 2>&1 | tail -5
DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 2 1 64 3 2>&1 | tail -12
DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 2 1 48 3 2>&1 | tail -3
