cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /tmp/b.log 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
head -1 $f
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def isf(n): return ('Cijk' in n) or ('chol_inv' in n) or ('block_copy' in n)
# find refactor windows: contiguous runs of factor kernels
runs=[];cur=[]
for r in rows:
    if isf(r['Kernel_Name']): cur.append(r)
    else:
        if len(cur)>50: runs.append(cur)
        cur=[]
if len(cur)>50: runs.append(cur)
print("refactors",len(runs))
run=runs[-1]
t0=int(run[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in run)
print("wall_us",(t1-t0)/1e3,"kernels",len(run))
byq=collections.defaultdict(list)
for r in run: byq[r['Queue_Id']].append(r)
for q,rs in byq.items():
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rs)/1e3
    gaps=[(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3 for a,b in zip(rs,rs[1:])]
    print("queue",q,"n",len(rs),"busy_us %.0f"%busy,"span_us %.0f"%((int(rs[-1]['End_Timestamp'])-int(rs[0]['Start_Timestamp']))/1e3),"gaps: n>2us",sum(g>2 for g in gaps),"sum_gap_us %.0f"%sum(g for g in gaps if g>0), "median %.2f"%sorted(gaps)[len(gaps)//2])
for q,rs in byq.items():
    segs=[];st=int(rs[0]['Start_Timestamp']);en=int(rs[0]['End_Timestamp'])
    for r in rs[1:]:
        if int(r['Start_Timestamp'])-en>5000: segs.append((st,en)); st=int(r['Start_Timestamp'])
        en=int(r['End_Timestamp'])
    segs.append((st,en))
    print("queue",q,"busy segments (us):",["%.0f-%.0f"%((a-t0)/1e3,(b-t0)/1e3) for a,b in segs])
# timeline of the main queue, coarse
mainq=max(byq,key=lambda q:len(byq[q]))
for r in byq[mainq][:0]:
    n=r['Kernel_Name']; n='node128' if 'node128' in n else ('base' if 'base' in n else ('copy' if 'block_copy' in n else 'gemm'))
    print("%8.1f %8.1f %s"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,n))
PY
