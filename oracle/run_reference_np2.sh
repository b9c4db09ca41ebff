#!/bin/bash
# oracle/run_reference_np2.sh -- TEST INFRASTRUCTURE ONLY (build container; needs /root/reference and oracle/_ref).
# Runs the real reference on 2 MPI ranks (process grid 2x1x1x1; argument 4: 4 ranks, 2x2x1x1) on its own 8^4 sample
# configuration with the sample.ini hierarchy and prints the log from which tests/golden/ref_8x8_3lvl_np2.json
# (ref_8x8_3lvl_np4.json) was taken.
set -e
NP=${1:-2}
if [ "$NP" = 4 ]; then L0="4 4 8 8"; L1="2 2 4 4"; L2="1 1 2 2"; else L0="4 8 8 8"; L1="2 4 4 4"; L2="1 2 2 2"; fi
HERE=$(cd "$(dirname "$0")" && pwd)
TMP=$(mktemp -d)
cat > $TMP/np2.ini <<EOF
configuration: ${DDAMG_REFERENCE:-/root/reference}/conf/8x8x8x8b6.0000id3n1
format: 0
right hand side: 0
antiperiodic boundary conditions: 1
number of levels: 3
number of openmp threads: 1
d0 global lattice: 8 8 8 8
d0 local lattice: $L0
d0 block lattice: 2 2 2 2
d0 post smooth iter: 2
d0 block iter: 4
d0 test vectors: 28
d0 setup iter: 4
d1 global lattice: 4 4 4 4
d1 local lattice: $L1
d1 block lattice: 2 2 2 2
d1 post smooth iter: 2
d1 block iter: 4
d1 test vectors: 28
d1 setup iter: 3
d2 global lattice: 2 2 2 2
d2 local lattice: $L2
m0: -0.5
csw: 1.0
tolerance for relative residual: 1E-10
iterations between restarts: 50
maximum of restarts: 20
coarse grid tolerance: 5E-2
coarse grid iterations: 100
coarse grid restarts: 5
print mode: 1
method: 2
mixed precision: 1
randomize test vectors: 0
EOF
cd $TMP && /opt/conda/bin/mpiexec -n $NP $HERE/_ref/dd_alpha_amg_scalar np2.ini
