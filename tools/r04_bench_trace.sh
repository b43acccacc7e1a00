# the kernel sequence of bench.py's steps with the gaps between kernels (rocprofv3 --kernel-trace): where a step's wall time
# goes beyond its two big kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/benchtr; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 --no-collect > $O/log.txt 2>&1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/benchtr/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the timed region: the last 4*... take the kernels between the 3rd-last and last k_encode_stream
idx=[i for i,r in enumerate(rows) if 'k_encode_stream' in r['Kernel_Name']]
a=idx[-8] if len(idx)>=8 else idx[0]
t0=None
for r in rows[a:a+40]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-t0)/1000 if t0 else 0
    print('%-58s gap %8.1f us  dur %9.1f us' % (r['Kernel_Name'][:58], gap, (e-s)/1000))
    t0=e
PY
