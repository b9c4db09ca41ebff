cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/se1
for args in "" "--self-exchange=-1,-1,-1,1" "--self-exchange=-1,-1,-1,1 --gather 1" "--levels 3" "--levels 3 --self-exchange=-1,-1,-1,1" "--levels 3 --self-exchange=-1,-1,-1,1 --gather 1"; do
  echo "== $args"; python3 tools/solve_bench.py --lattice 32 32 32 32 --solves 3 $args 2>&1 | grep -E "solve_s|rror" | tail -1
done
