# launch sequence (start offset, duration) of the Schwarz / transport kernels of the rehearsed 8-GPU solve: do exchanges overlap?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rseq; mkdir -p $O
rocprofv3 --kernel-trace -d $O/t -o t -- python3 tools/rehearse_profile.py 8 1 2 > $O/solve.log 2>$O/err.log
python3 - $O/t/t_results.db <<'PY'
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select start, end, name from kernels order by start"))
rows = rows[-1500:]
# first smoother call in this window: print 40 kernels from the first sap_face_pack
i0 = next(i for i, r in enumerate(rows) if 'sap_face_pack' in r[2])
t0 = rows[i0][0]
for s, e, name in rows[i0:i0 + 34]:
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("ddamg::", ""); short = re.sub(r"<.*", "", short)
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f})  {short}")
PY
rm -rf $O/t
