# diagnostic: what the couplings across tile faces cost in the fine operator, per direction.
# Needs fine_op.hip compiled with -DDDAMG_FACE_DIAG (hipcc ... -DDDAMG_FACE_DIAG -c fine_op.hip -o build/fine_op.o, relink);
# DDAMG_FACE_MASK: bit mu = forward, bit 4+mu = backward coupling across the tile face is computed (0xff: all).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in ${MASKS:-0xff 0x00 0x0f 0xf0 0xfe 0xfd 0xfb 0xf7 0xef 0xdf 0xbf 0x7f}; do
  echo "mask $m: $(DDAMG_FACE_MASK=$m python3 bench.py --steps 300 --warmup 50 --no-solve --no-strong --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1000,2))')"
done
