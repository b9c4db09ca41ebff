# wall-clock seconds per setup phase (DDAMG_SETUP_TIMING); arguments: "L levels" pairs, default 32 2 / 48 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/setup_timing
for cfg in "${@:-32 2}"; do set -- $cfg
  DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 1 1 $1 $2 > gpurun_out/setup_timing/s$1.log 2>&1
  grep -i -v "^W2026" gpurun_out/setup_timing/s$1.log | tail -12
done
