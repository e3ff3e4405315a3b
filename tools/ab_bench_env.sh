# bench.py under several environment settings alternating on ONE box:
#   bash tools/ab_bench_env.sh <rounds> "VAR=a" "VAR=b VAR2=c" ... [-- bench.py arguments]      ("-" = defaults)
# prints per run: ms per step, iterations, back-solve launch, per-step breakdown of the bench workload (AB_EXTRA: extra workloads)
R=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" = "--" ] && shift
for r in $(seq 1 $R); do
  for e in "${envs[@]}"; do
    ee="$e"; [ "$e" = "-" ] && ee=""
    env $ee python bench.py --no-cpu-baseline --extra-workloads "${AB_EXTRA:-}" "$@" 2>/dev/null | tail -1 | E="$e" python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
w=d.get('workloads',[])
print('[%s]'%os.environ['E'], 'ms/step', d['ms_per_step'], 'it', d.get('iters_per_frame'), 'bs_ms', d['roofline'].get('avg_launch_ms'), d.get('step_breakdown_ms'), ' | '.join('%s %.3f ms it %.1f'%(x.get('name',x.get('workload')), x['ms_per_step'], x.get('iters_per_frame',0)) for x in w[1:]))
"
  done
done
