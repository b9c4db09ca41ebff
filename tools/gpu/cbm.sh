# level-1 Schwarz block solver: workgroups per CU (blocks in flight against the Infinity Cache)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 0 1 2; do
  echo "wg_per_cu $w: $(DDAMG_COARSE_SAP_WG_PER_CU=$w python3 tools/solve_profile.py 3 1 64 3 2>&1 | tail -1 | cut -c1-110)"
done
