cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb -- python /root/repo/tools/run_gpu.py synbar:160x36x36:512 3 > /tmp/bb.log 2>&1
tail -3 /tmp/bb.log
f=$(find /tmp/profb -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    n=r['Name']; n=n if len(n)<60 else n[:28]+'..'+n[-28:]
    print(f"{n:60s} calls {int(r['Calls']):5d} tot_ms {float(r['TotalDurationNs'])/1e6:8.2f} avg_us {float(r['AverageNs'])/1e3:8.1f}  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
