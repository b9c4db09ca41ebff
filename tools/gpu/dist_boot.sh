# bootstrap with shared passes over P on a process grid: the rehearsed 8-GPU setup, both ways, twice (the first process on a
# fresh box pays one-time costs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/rehearse_profile.py 8 1 1 > /dev/null 2>&1
for rep in 1 2; do
echo "single-process-only: $(DDAMG_BOOTSTRAP_BATCHED_SINGLE_PROCESS_ONLY=1 python3 tools/rehearse_profile.py 8 1 2 | tail -1 | cut -c1-130)"
echo "batched: $(python3 tools/rehearse_profile.py 8 1 2 | tail -1 | cut -c1-130)"
done
