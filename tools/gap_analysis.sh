cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/b.log 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def short(n):
    for k in ['loop_control','vertex_gather','elem_energy','build_p','build_q','spmv','merge','step_forward','reduce_partial','backsolve','elem_hessian','assemble','dense_fill','chol_inv','block_copy','Cijk','fillBuffer','be_update','init_x']:
        if k in n: return k
    return n[:20]
gaps=collections.defaultdict(list)
prev=None
for r in rows:
    n=short(r['Kernel_Name']); s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if prev: gaps[(prev[0],n)].append(s-prev[1])
    prev=(n,e)
for k,v in sorted(gaps.items(), key=lambda kv:-sum(kv[1])):
    if len(v)>=50: print(f"{k[0]:>16s} -> {k[1]:<16s} n {len(v):5d} avg_gap_us {sum(v)/len(v)/1e3:8.2f} total_ms {sum(v)/1e6:8.2f}")
PY
