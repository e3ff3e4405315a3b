# per-kernel time table of a workload under an environment (tools/run_case.py honours RUN_CASE_FLAGS):
#   bash tools/prof_env.sh <workload> <steps> <tag> "VAR=a VAR2=b"     (GPU box; writes gpurun_out/<tag>_kstats.csv)
wl=${1:-bar17K_twist}; steps=${2:-10}; tag=${3:-prof}; e=${4:-}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
env $e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python /root/repo/tools/run_case.py $wl - $steps > /tmp/prof_$tag.log 2>&1
tail -2 /tmp/prof_$tag.log
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
mkdir -p /root/repo/gpurun_out; cp $f /root/repo/gpurun_out/${tag}_kstats.csv
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    n=r['Name']; n=n if len(n)<64 else n[:30]+'..'+n[-30:]
    print(f"{n:64s} calls {int(r['Calls']):6d} tot_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:8.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
