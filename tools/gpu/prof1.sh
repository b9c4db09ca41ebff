set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof1
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof1/s32 -o s32 -- python3 tools/solve_profile.py 10 1 32 2 > gpurun_out/prof1/s32.log 2>&1
python3 tools/rocpd_export.py stats gpurun_out/prof1/s32/s32_results.db gpurun_out/prof1/s32_stats.csv
tail -2 gpurun_out/prof1/s32.log | head -1; head -25 gpurun_out/prof1/s32_stats.csv
python3 tools/gpu_busy.py gpurun_out/prof1/s32/s32_results.db 0.45
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof1/s48 -o s48 -- python3 tools/solve_profile.py 5 1 48 3 > gpurun_out/prof1/s48.log 2>&1
python3 tools/rocpd_export.py stats gpurun_out/prof1/s48/s48_results.db gpurun_out/prof1/s48_stats.csv
grep solve_s gpurun_out/prof1/s48.log; head -30 gpurun_out/prof1/s48_stats.csv
python3 tools/gpu_busy.py gpurun_out/prof1/s48/s48_results.db 1.0
