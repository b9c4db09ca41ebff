#!/bin/bash
# tools/gpu/run.sh -- the ONE launcher for everything that is run on the one-GPU MI355X box:
#     gpurun -- 'bash tools/gpu/run.sh <recipe> [arguments]'
# Output goes under gpurun_out/<recipe>/.  The recipes are built from four building blocks (tests, phases, stats, pmc) and two
# ways of comparing on ONE box (env: a knob on / off in alternation; lib: another build of the library through DDAMG_HIP_LIBRARY).
# `bash tools/gpu/run.sh help` lists them.  (Rounds 1-3 kept one script per experiment here: 49 of them; git history has them.)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
R=${1:-help}; shift
O=gpurun_out/$R; mkdir -p "$O"
SP="python3 tools/solve_profile.py"      # N_solves mixed_precision extent levels

warm() { $SP 1 1 32 2 > /dev/null 2>&1; }   # the first process on a fresh box pays one-time allocation costs (DESIGN, setup)
# kernel statistics of a command: stats <label> <command...>  ->  $O/<label>_stats.csv
stats() { local l=$1; shift; rocprofv3 --kernel-trace --stats -d $O/kt_$l -o k -- "$@" > $O/$l.log 2>&1; python3 tools/rocpd_export.py stats $O/kt_$l/k_results.db $O/${l}_stats.csv; rm -rf $O/kt_$l; }
# counter passes of a command, one rocprofv3 run per quoted group: pmc <label> "<counters>" ["<counters>" ...] -- <command...>  ->  $O/<label>.json
pmc() { local l=$1 i=0 dbs=""; shift; local groups=(); while [ "$1" != "--" ]; do groups+=("$1"); shift; done; shift
  for c in "${groups[@]}"; do i=$((i + 1)); timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${l}_$i -o p -- "$@" > /dev/null 2>> $O/$l.err || echo "pass $i ($c) failed"; dbs="$dbs $O/pmc_${l}_$i/p_results.db"; done
  python3 tools/rocpd_export.py pmc $dbs > $O/$l.json; rm -rf $O/pmc_${l}_*; }
# setup phase times and the solve of one extent: phases <extent> <levels> [label]
phases() { DDAMG_SETUP_TIMING=1 $SP ${4:-1} 1 $1 $2 > $O/phases_${3:-$1}.log 2>&1; grep -E "ddamg setup|lattice" $O/phases_${3:-$1}.log | cut -c1-230; }
short() { sed 's/(HIP_vector[^"]*"/"/; s/(float[^"]*"/"/; s/(ddamg::[^"]*"/"/; s/void ddamg:://; s/(anonymous namespace):://' | cut -c1-150; }

case $R in
help) grep -E "^[a-z0-9_|]+\)" "$ROOT/tools/gpu/run.sh" | sed 's/)\s*#/  --/; s/)$//' ;;

tests)        # smoke() + the -m gpu suite (arguments: extra pytest arguments, e.g. tests/test_gpu_three_levels.py -k smoother)
  python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
  timeout -k 10 1100 python3 -m pytest ${@:-tests/} -x -q -m gpu --durations=6 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests.log ;;
tests_full_links)   # the suite on the full link storage (the path every non-SU(3) field takes)
  DDAMG_LINK_COMPRESSION=0 timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_dirac.py::test_two_row_link_storage_and_its_fall_back > $O/tests.log 2>&1; echo "rc=$?"; tail -4 $O/tests.log ;;
bands)        # iteration counts of the full-size tests, printed (what the +-1 assertions are pinned to)
  timeout -k 10 1100 python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_configs.py tests/test_gpu_reference_volumes.py -q -m gpu -s 2>&1 | grep -E "level|method|passed|failed|Error|assert|iterations|residual curve" | tee $O/bands.log | tail -30 ;;
phases)       # wall-clock seconds per setup phase: run.sh phases "32 2" "48 3" "64 3" (a warm-up process first)
  warm; for cfg in "${@:-32 2}"; do set -- $cfg; echo "== $1^4, $2 levels"; phases $1 $2; done ;;
env)          # a knob on / off in alternation on one box: run.sh env "DDAMG_X=1 DDAMG_Y=2" <extent> <levels> [repetitions]  (setup phases + solve)
  K=$1; E=${2:-32}; L=${3:-2}; warm
  for rep in $(seq ${4:-2}); do echo "== default"; phases $E $L def$rep; echo "== $K"; ( export $K; phases $E $L knob$rep ); done ;;
lib)          # another build of the library against the current one: run.sh lib <path/to/libddamg_hip_base.so> sap|"<extent> <levels>"
  B=$(realpath $1); shift; warm
  for rep in 1 2; do for lib in base new; do
    unset DDAMG_HIP_LIBRARY; [ $lib = base ] && export DDAMG_HIP_LIBRARY=$B
    if [ "$1" = sap ]; then echo "$lib:"; SAP_BENCH_ITERS=${SAP_AB_ITERS:-0,4} python3 tools/sap_bench.py 2>&1 | grep block_iter
    else echo "$lib: $($SP 3 1 $1 2>&1 | tail -1 | cut -c1-170)"; fi      # "$1" = "<extent> <levels>", split by the shell here
  done; done ;;
stats)        # kernel statistics of setup + N solves: run.sh stats <extent> <levels> [N] [grep pattern]
  stats s$1 $SP ${3:-2} 1 $1 $2; tail -1 $O/s$1.log | cut -c1-200; if [ -n "$4" ]; then grep -E "$4" $O/s$1_stats.csv | short; else head -30 $O/s$1_stats.csv | short; fi ;;
mfma)         # matrix-core busy counters of a setup (many-right-hand-side kernels): run.sh mfma <extent> <levels>
  pmc mfma$1 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" -- $SP 0 1 $1 $2
  python3 - $O/mfma$1.json <<'PY'
import json, sys
for k, v in json.load(open(sys.argv[1])).items():
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean", 0) > 0:
        b = v["SQ_VALU_MFMA_BUSY_CYCLES"]; cyc = v["GRBM_GUI_ACTIVE"]["mean"] / 8
        print(f"{k[:72]:72s} busy {b['mean'] / (cyc * 1024):.3f}  launches {b['launches']}")
PY
  ;;
timeline)     # kernel timeline of the last solves with the idle gaps: run.sh timeline <extent> <levels> [kernels]
  rocprofv3 --kernel-trace -d $O/t -o t -- $SP 3 1 ${1:-32} ${2:-2} > $O/solve.log 2>$O/err.log
  python3 tools/kernel_timeline.py $O/t/t_results.db ${3:-1500} > $O/timeline.txt; rm -rf $O/t; grep "^#" $O/timeline.txt | head -40; tail -1 $O/solve.log ;;
sap)          # Schwarz kernel: smoother parity tests, then time per smoother call for block_iter 0 and 4 (fixed part / MinRes steps)
  timeout -k 10 600 python3 -m pytest tests/test_gpu_multigrid.py -x -q -m gpu -k "smoother" 2>&1 | tail -2 && SAP_BENCH_ITERS=${1:-0,4} python3 tools/sap_bench.py 2>&1 | grep block_iter ;;
sap_chain)    # the MinRes step of the Schwarz kernel segment by segment (needs `make -C ddalphaamg_amd/csrc diag`) -> profiles/r04_sap_chain.md
  DDAMG_HIP_LIBRARY=$PWD/ddalphaamg_amd/libddamg_hip_diag.so python3 tools/sap_chain.py ${1:-4} > $O/sap_chain.md 2> $O/err.log; cat $O/sap_chain.md; tail -3 $O/err.log
  echo "--- the same call on the product build:"; SAP_BENCH_ITERS=${1:-4} python3 tools/sap_bench.py 2>&1 | grep block_iter ;;
sap_pmc)      # traffic and issue counters of the Schwarz kernel
  export SAP_BENCH_ITERS=4    # (never `env VAR=.. program` behind rocprofv3's `--`: the profiler has initialised the GPU, and that hop is an exec)
  pmc sap FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" -- python3 tools/sap_bench.py ;;
transfer_context)  # restriction / interpolation alone: back to back, then after other kernels that write (default) or only read 1 GB
  for ro in 0 1; do
    [ $ro = 1 ] && export TRANSFER_BENCH_READ_ONLY=1
    TRANSFER_BENCH_INTERLEAVE=1 rocprofv3 --kernel-trace -d $O/t$ro -o t -- python3 tools/transfer_bench.py > $O/tb$ro.log 2>$O/err$ro.log
    echo "== other traffic read-only: $ro   (23 launches back to back, then 10 with the other traffic in between; us)"
    KSEQ_PERIOD=1 python3 tools/kernel_seq.py $O/t$ro/t_results.db "restrict_kernel<float, 1>" 33; KSEQ_PERIOD=1 python3 tools/kernel_seq.py $O/t$ro/t_results.db "interpolate_kernel<float>" 33
    rm -rf $O/t$ro
  done ;;
transfer_pmc) # restriction against interpolation (same bytes): issue, wait, LDS and cache counters of a 32^4 solve
  pmc transfer FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
      "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" -- $SP 5 1 32 2
  python3 - $O/transfer.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in d:
    if k.startswith("ddamg::restrict_kernel") or k.startswith("ddamg::interpolate_kernel"):
        print(k[:70]); print("   ", {c: round(v["mean"]) for c, v in d[k].items()})
PY
  ;;
selfx)        # fine operator through the RCCL self-exchange: 0, 1, 2, 3 split directions
  for g in "1,1,1,1" "-1,1,1,1" "-1,-1,1,1" "-1,-1,-1,1"; do
    echo "grid $g: $(python3 bench.py --steps 500 --warmup 100 --no-solve --no-strong --no-cpu-baseline --self-exchange=$g 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1000,2), "us")')"
  done ;;
rehearse)     # bench.py --rehearse N (default 8): the per-GPU problem of the N-GPU point on one GPU; run.sh rehearse "2 4 8" for the curve
  for n in ${1:-8}; do python3 bench.py --steps 50 --warmup 10 --no-solve --no-cpu-baseline --rehearse $n 2> $O/err$n.log | tail -1 > $O/line$n.json
    python3 -c "
import sys,json; d=json.loads(open('$O/line$n.json').read()); h=d['rehearsal']; s=d['strong_scaling']
print('N', h.get('n_gpus_rehearsed'), 'local', h.get('local_lattice'), 'n1', round(s['seconds_per_solve'],4), 'per gpu', round(h['seconds_per_solve_per_gpu'],4), 'plain', round(h['same_lattice_without_the_machinery']['seconds_per_solve'],4), 'cost of the machinery', h.get('cost_of_the_machinery'), 'predicted', round(h['predicted_seconds_per_solve_per_gpu'],4), 'speedup', round(h['predicted_speedup_vs_n1'],2), 'its', h['iterations'], 'setup', round(h['setup_seconds'],2), 'err', h.get('error'))"
  done ;;
rehearse_stats)  # kernel totals of the rehearsed N-GPU solve, through the machinery against the plain periodic lattice: run.sh rehearse_stats [N]
  for sx in 1 0; do stats r$sx python3 tools/rehearse_profile.py ${1:-8} $sx 2; tail -1 $O/r$sx.log | cut -c1-170; head -16 $O/r${sx}_stats.csv | short; done ;;
driver_line)  # the driver's invocation of bench.py, timed, as the first process of the box
  t0=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/line.json 2> $O/err.log; echo "rc=$? wall $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s"
  python3 -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','vs_baseline','dtype','data')}); print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline']['kind'], d['cpu_baseline']['cores'])
for leg in ('solve','three_level_48','strong_scaling'):
    if leg in d: print(leg, {k:v for k,v in d[leg].items() if k not in ('workload','coarse_operator','reference_32','per_level_messages')})" ;;
profile)      # every artefact of profiles/<round>_* : run.sh profile r04 ; then python3 tools/commit_profiles.py gpurun_out/profile r04
  RD=${1:-r04}; warm
  rocprofv3 --kernel-trace --stats -d $O/bench -o bench -- python3 bench.py > $O/bench_line.json 2> $O/bench.err
  python3 tools/rocpd_export.py stats $O/bench/bench_results.db $O/${RD}_bench_kernel_stats.csv; rm -rf $O/bench; tail -c 400 $O/bench_line.json; echo
  pmc ${RD}_pmc_bench FETCH_SIZE WRITE_SIZE -- python3 bench.py --steps 25 --warmup 5 --no-solve --no-strong --no-cpu-baseline
  export SAP_BENCH_ITERS=4
  pmc ${RD}_pmc_sap FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" -- python3 tools/sap_bench.py
  unset SAP_BENCH_ITERS
  for cfg in "32 2 10" "48 3 5" "64 3 3"; do set -- $cfg; stats ${RD}_solve$1 $SP $3 1 $1 $2; mv $O/${RD}_solve$1_stats.csv $O/${RD}_solve$1_kernel_stats.csv; tail -1 $O/${RD}_solve$1.log | cut -c1-200; done
  pmc ${RD}_pmc_mfma_lockstep32 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" -- $SP 1 1 32 2
  pmc ${RD}_pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" -- $SP 1 1 48 3
  pmc ${RD}_pmc_mfma64 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" -- $SP 0 1 64 3
  stats ${RD}_mass_shift python3 tools/mass_shift_trace.py; mv $O/${RD}_mass_shift_stats.csv $O/${RD}_mass_shift_kernel_stats.csv
  ls $O ;;
*) echo "unknown recipe $R"; bash "$ROOT/tools/gpu/run.sh" help; exit 2 ;;
esac
