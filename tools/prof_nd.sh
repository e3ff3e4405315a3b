cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/b.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:32]:
    n=r['Name']; n=n if len(n)<70 else n[:34]+'..'+n[-30:]
    print(f"{n:70s} calls {int(r['Calls']):5d} tot_ms {float(r['TotalDurationNs'])/1e6:8.2f} avg_us {float(r['AverageNs'])/1e3:8.1f}  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
cp $f /root/repo/gpurun_out/nd_kernel_stats.csv
tail -1 /tmp/b.log | cut -c1-400
