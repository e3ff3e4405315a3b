python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed"
for cfg in "2 768 1 1" "3 384 1 1"; do set -- $cfg
echo "== levels $1 min $2 parallel $3 graph $4"
DOTMI_ND_LEVELS=$1 DOTMI_ND_MIN=$2 DOTMI_ND_PARALLEL=$3 DOTMI_FACTOR_GRAPH=$4 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], d['step_breakdown_ms'], r['avg_launch_ms'], r['algorithmic_bytes_per_launch'], r['frac'], d['iters_per_frame'])
"
done
