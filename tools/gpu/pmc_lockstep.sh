#!/bin/bash
# issue / wait counters of the lockstep matrix-core kernels (32^4 two-level setup): why ls_hop_kernel keeps the matrix cores busy 0.31
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_ls; mkdir -p $O
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $O/p$i -o p -- python3 tools/solve_profile.py 0 1 32 2 > /dev/null 2>> $O/err.log || echo "pass $i ($c) failed"
done
python3 tools/rocpd_export.py pmc $O/p*/p_results.db > $O/pmc_lockstep.json 2>> $O/err.log
rm -rf $O/p*/
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_ls/pmc_lockstep.json'))
for k in d:
    if 'ls_hop' in k or 'ls_self' in k or 'coarse_site_kernel<float, 6, 1>' in k:
        print(k[:60]); print({c:(round(v['mean']), v['launches']) for c,v in d[k].items()})
PY
