# average duration of the loop kernels under rocprofv3 for several environment settings (one bench run each)
#   bash tools/prof_kernels_env.sh "VAR=a" "VAR=b VAR2=c" ...    ("-" = defaults)
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  rm -rf /tmp/prof_pk
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pk -- python /root/repo/bench.py --no-cpu-baseline --extra-workloads none --steps ${PK_STEPS:-8} --warmup 2 ${PK_ARGS} > /tmp/prof_pk.log 2>&1
  f=$(find /tmp/prof_pk -name "*kernel_stats.csv" | head -1)
  echo "== [$e]"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "dotmi::" in n and int(r["Calls"]) > 100:
        short = n.split("dotmi::")[1].split("(")[0]
        print(f"   {short:44s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
  grep '^{' /tmp/prof_pk.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ms/step', d['value'], d['step_breakdown_ms'])"
done
