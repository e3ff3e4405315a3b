# per-kernel times of the two-level back-solve (rocprofv3 kernel trace) -- usage: bash tools/tl_prof.sh <workload> "<TL_ENV>" [steps]
W=${1:-synbar:140x35x35:256}; E=${2:-}; S=${3:-2}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlprof
TL_ONLY=1 TL_ENV="$E" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tlprof -- python /root/repo/tools/tl_check.py "$W" $S > /tmp/tlprof.log 2>&1
grep -v "^W2026\|^\[" /tmp/tlprof.log | tail -8
f=$(find /tmp/tlprof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us {float(r['TotalDurationNs'])/tot*100:6.2f}%")
PY
