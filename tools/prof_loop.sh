# per-kernel average durations of the L-BFGS loop kernels under rocprofv3 for the current environment
# usage (GPU box): bash tools/prof_loop.sh <tag> [bench args]
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python /root/repo/bench.py --no-cpu-baseline --extra-workloads none "$@" > /tmp/prof_$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
mkdir -p /root/repo/gpurun_out
cp $f /root/repo/gpurun_out/${tag}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "dotmi::" in n:
        short = n.split("dotmi::")[1].split("(")[0]
        print(f"{short:32s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.2f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
grep '^{' /tmp/prof_$tag.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['step_breakdown_ms'])"
