cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nab1tr; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 200 rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python3 tools/workload.py nab1 --steps 3 > $O/log.txt 2>&1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/nab1tr/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 30 kernels
t0=None
for r in rows[-26:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap = (s-t0)/1000 if t0 else 0
    print('%-60s start+%7.1f us gap  dur %7.1f us' % (r['Kernel_Name'][:60], gap, (e-s)/1000))
    t0=e
PY
