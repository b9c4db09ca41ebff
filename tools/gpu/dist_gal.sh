# batched coarse Galerkin construction on a process grid: the rehearsed 8-GPU setup, batched against column by column
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/rehearse_profile.py 8 1 1 > /dev/null 2>&1
for rep in 1 2 3; do
echo "batched: $(python3 tools/rehearse_profile.py 8 1 2 | tail -1 | cut -c55-130)"
echo "column by column: $(DDAMG_COARSE_GALERKIN_DIST_UNBATCHED=1 python3 tools/rehearse_profile.py 8 1 2 | tail -1 | cut -c55-130)"
done
echo "plain: $(python3 tools/rehearse_profile.py 8 0 2 | tail -1 | cut -c55-130)"
