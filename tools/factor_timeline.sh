# timeline of ONE refactorisation (last of a short bench run): start us, duration us, queue, kernel class, workgroups
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --extra-workloads none "$@" > /tmp/b.log 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def cls(n):
    if 'tile_gemm' in n: return 'tgemm'
    if 'tile_task' in n: return 'tile'
    if 'clear_tiles' in n: return 'clear'
    if 'node128' in n: return 'node128'
    if 'chol_inv_base' in n: return 'base64'
    if 'block_copy' in n: return 'copy'
    if 'Cijk' in n: return 'gemm'
    if 'elem_hessian' in n: return 'elemH'
    if 'assemble' in n: return 'assemble'
    if 'clear_segments' in n: return 'clear'
    if 'dense_fill' in n: return 'fill'
    if 'pad_identity' in n: return 'pad'
    return None
runs=[];cur=[]
for r in rows:
    c=cls(r['Kernel_Name'])
    if c: cur.append(r)
    else:
        if len(cur)>20: runs.append(cur)
        cur=[]
if len(cur)>20: runs.append(cur)
fr=[r for r in runs if any(cls(k['Kernel_Name']) in ('tile','tgemm','gemm') for k in r)]
run=fr[-1]
t0=int(run[0]['Start_Timestamp'])
qs={}
for r in run:
    q=r['Queue_Id']; qs.setdefault(q,len(qs))
    wg=int(r['Grid_Size_X'])*int(r['Grid_Size_Y'])*int(r['Grid_Size_Z'])//max(1,int(r['Workgroup_Size_X'])*int(r['Workgroup_Size_Y'])*int(r['Workgroup_Size_Z']))
    print("%8.1f %7.1f q%d %-8s wg %5d"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,qs[q],cls(r['Kernel_Name']),wg))
print("wall", (max(int(r['End_Timestamp']) for r in run)-t0)/1e3)
PY
