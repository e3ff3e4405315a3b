timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E" | head -20
for sr in 1 0; do
echo "== split root $sr"
DOTMI_ND_SPLIT_ROOT=$sr timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], d['step_breakdown_ms'], r['avg_launch_ms'], r['launches_timed'], r['frac'], d['iters_per_frame'])
"
done
