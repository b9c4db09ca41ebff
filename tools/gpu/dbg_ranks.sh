cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1
for i in 1 2 3 4 5 6; do
python3 bench.py --gpus 2 --lattice 16 16 16 16 --strong-lattice 16 16 16 16 --steps 5 --warmup 2 --transport host --no-cpu-baseline > gpurun_out/dbg_ranks.out 2> gpurun_out/dbg_ranks.err; rc=$?; echo "run $i rc=$rc"
if [ $rc != 0 ]; then grep -v "amdgpu.ids\|c10d" gpurun_out/dbg_ranks.err | tail -40; break; fi
done
