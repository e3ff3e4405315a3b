# rocprofv3 per-kernel averages of a short bench run of ONE workload under several environment settings
#   bash tools/prof_workload_env.sh <workload> <steps> "VAR=a" "-" ...
wl=$1; steps=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  rm -rf /tmp/prof_pw
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pw -- python /root/repo/bench.py --workload $wl --no-cpu-baseline --extra-workloads none --steps $steps --warmup 2 > /tmp/prof_pw.log 2>&1
  f=$(find /tmp/prof_pw -name "*kernel_stats.csv" | head -1)
  echo "== $wl [$e]"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dotmi::" in n and float(r["Percentage"]) > 0.8:
        short = n.split("dotmi::")[1].split("(")[0]
        print(f"   {short:44s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.2f} us  {float(r['Percentage']):5.1f} %")
PY
  grep '^{' /tmp/prof_pw.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ms/step', d['value'], d['step_breakdown_ms'])"
done
