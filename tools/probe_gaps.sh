export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/probe_gaps -o g -- python3 tools/probe_gaps.py > gpurun_out/probe_gaps.log 2>&1 || exit 1
python3 - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/probe_gaps/g_kernel_trace.csv'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None; last=None; acc={}
for r in rows:
    a,b=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    k=r['Kernel_Name'][:40]+' lds='+r.get('LDS_Block_Size','?')
    if prev is not None and last==k:
        acc.setdefault(k,[]).append(((a-prev)/1e3,(b-a)/1e3))
    prev=b; last=k
for k,v in acc.items():
    g=sorted(x[0] for x in v); d=sorted(x[1] for x in v)
    print('%-60s n=%3d gap median %5.1f us  dur median %6.1f us' % (k,len(v),g[len(g)//2],d[len(d)//2]))
PY
