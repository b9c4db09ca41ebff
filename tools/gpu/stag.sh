cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in "0 0" "1 4" "1 7" "1 10" "2 7" "3 7"; do set -- $m
  echo "== stagger $1 sleeps $2: $(DDAMG_SAP_STAGGER=$1 DDAMG_SAP_STAGGER_SLEEPS=$2 SAP_BENCH_ITERS=4 python3 tools/sap_bench.py 2>&1 | grep block_iter)"
done
