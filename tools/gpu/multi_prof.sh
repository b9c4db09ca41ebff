#!/bin/bash
# kernel statistics and matrix-core busy counters of a three-level setup (the many-right-hand-side kernels of the intermediate level);
# argument: extent (default 48)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
E=${1:-48}; O=gpurun_out/multi_prof; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python3 tools/solve_profile.py 0 1 $E 3 > $O/run.log 2>&1
python3 tools/rocpd_export.py stats $O/kt/k_results.db $O/stats$E.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc -o p -- python3 tools/solve_profile.py 0 1 $E 3 > /dev/null 2>> $O/run.log
python3 tools/rocpd_export.py pmc $O/pmc/p_results.db > $O/pmc_mfma$E.json
rm -rf $O/kt $O/pmc
head -25 $O/stats$E.csv | cut -c1-90,150-330
