# Galerkin phase against the slab size of the batched construction
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "32 2" "48 3" "64 3"; do set -- $cfg
  for sl in 0 64 512 4096; do
    if [ $sl = 0 ]; then unset DDAMG_GALERKIN_SLAB_AGGS; else export DDAMG_GALERKIN_SLAB_AGGS=$sl; fi
    echo "lattice $1 slab $sl: $(DDAMG_SETUP_TIMING=1 python3 tools/solve_profile.py 1 1 $1 $2 2>&1 | grep -E 'Galerkin|setup_s' | sed 's/.*Galerkin coarse operator//; s/.*"setup_s": \([0-9.]*\).*/setup \1/' | tr '\n' ' ')"
  done
done
