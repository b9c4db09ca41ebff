# diagnostic (build fine_op.hip with -DDDAMG_FACE_DIAG): time of the fine operator with the couplings that leave a tile
# switched off per direction (DDAMG_FACE_MASK) and with the load variants (DDAMG_DIRAC_OPT)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in ${OPTS:-0 1 2 3}; do for m in ${MASKS:-0xff 0x00}; do
  echo "opt $o mask $m: $(DDAMG_DIRAC_OPT=$o DDAMG_FACE_MASK=$m python3 bench.py --steps 300 --warmup 50 --no-solve --no-strong --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1000,2))')"
done; done
