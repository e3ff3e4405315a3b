# bench.py with two builds of the library alternating on ONE box (box-to-box noise is +-3 %, more than most changes):
#   bash tools/ab_lib.sh <libA.so> <libB.so> [rounds] [bench.py arguments ...]
# prints per run: ms_per_step, iterations, loop / factor ms of the bench workload (+ extra workloads if asked for)
A=$1; B=$2; R=${3:-3}; shift; shift; shift
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    DOTMI_LIBRARY=$(realpath $L) python bench.py --no-cpu-baseline --extra-workloads "${AB_EXTRA:-}" "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
w=d.get('workloads',[])
print('$L'.split('/')[-1], 'ms/step', d['ms_per_step'], 'it', d.get('iters_per_frame', d.get('config',{}).get('iters_per_frame')), 'bs_us', d['roofline'].get('avg_launch_us', d['roofline'].get('avg_launch_ms')), 'frac', d['roofline']['frac'], ' | '.join('%s %.3f ms it %.1f'%(x.get('name',x.get('workload')), x['ms_per_step'], x.get('iters_per_frame',0)) for x in w))
"
  done
done
