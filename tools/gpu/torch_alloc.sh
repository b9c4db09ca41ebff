# does the HIP runtime that torch brings along change the allocation cost of a 64^4 setup?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "plain:"; DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 1 1 64 3 2>&1 | grep -E "hierarchy|Galerkin|setup_s"
echo "torch first:"; DDAMG_IMPORT_TORCH=1 DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 1 1 64 3 2>&1 | grep -E "hierarchy|Galerkin|setup_s"
python3 -c "
import ctypes, os
for l in open('/proc/self/maps'): pass
import torch; print(torch.__file__); print([x for x in os.listdir(os.path.join(os.path.dirname(torch.__file__),'lib')) if 'hip' in x or 'hsa' in x or 'rccl' in x][:10])"
