cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r02; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  d=$(echo $c | tr ' ' '_' | cut -c1-40)
  SAP_BENCH_ITERS=4 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_sap_$d -o p -- python3 tools/sap_bench.py > /dev/null 2>> $O/bench.err
done
python3 tools/rocpd_export.py pmc $O/pmc_sap_*/p_results.db > $O/r02_pmc_sap.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_dirac_$c -o p -- python3 bench.py --steps 25 --warmup 5 --no-solve --no-strong --no-cpu-baseline > /dev/null 2>> $O/bench.err
done
python3 tools/rocpd_export.py pmc $O/pmc_dirac_FETCH_SIZE/p_results.db $O/pmc_dirac_WRITE_SIZE/p_results.db > $O/r02_pmc_bench.json
rm -rf $O/*/
python3 -c "
import json
d=json.load(open('$O/r02_pmc_sap.json'))
for k,v in d.items():
    if 'sap' in k: print(k[:60], {c:(round(x['mean']),x['launches']) for c,x in v.items()})
"
