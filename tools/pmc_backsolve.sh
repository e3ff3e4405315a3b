# HBM traffic of the back-solve kernels from the PMC counters: one rocprofv3 pass per counter,
# kernel trace only (MI355X_MICROARCH.md, HBM section).  Writes gpurun_out/pmc_backsolve.txt
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out
: > /root/repo/gpurun_out/pmc_backsolve.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python /root/repo/tools/bench_backsolve.py > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c >> /root/repo/gpurun_out/pmc_backsolve.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name']==sys.argv[2]:
        n=r['Kernel_Name']
        k='backsolve' if 'backsolve' in n else ('reduce_partial' if 'reduce_partial' in n else None)
        if k: acc[k].append(float(r['Counter_Value']))
for k,v in acc.items(): print(sys.argv[2], k, 'dispatches', len(v), 'avg', sum(v)/len(v))
PY
done
tail -2 /tmp/pmc_FETCH_SIZE.log >> /root/repo/gpurun_out/pmc_backsolve.txt
cat /root/repo/gpurun_out/pmc_backsolve.txt
