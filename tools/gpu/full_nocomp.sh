# the GPU suite on the full link storage (the path every non-SU(3) field takes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
DDAMG_LINK_COMPRESSION=0 timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_dirac.py::test_two_row_link_storage_and_its_fall_back > gpurun_out/full/tests_nocomp.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/full/tests_nocomp.log
