# the matrix-core passes of profile_r03.sh alone (after a change to coarse_lockstep.* / coarse_batch.hip / coarse_op.h)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O; R=r03
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma32 -o p -- python3 tools/solve_profile.py 1 1 32 2 > /dev/null 2>> $O/bench.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma48 -o p -- python3 tools/solve_profile.py 1 1 48 3 > /dev/null 2>> $O/bench.err
python3 tools/rocpd_export.py pmc $O/pmc_mfma32/p_results.db > $O/${R}_pmc_mfma_lockstep32.json
python3 tools/rocpd_export.py pmc $O/pmc_mfma48/p_results.db > $O/${R}_pmc_mfma.json
rm -rf $O/*/; ls $O
