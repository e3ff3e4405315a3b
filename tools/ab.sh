# A/B of two builds of libdotmi.so on the bench workload: bash tools/ab.sh <base.so> <pairs> [bench args]
base=$1; pairs=${2:-3}; shift; shift
cp dot_amd/libdotmi.so /tmp/ab_new.so
for k in $(seq $pairs); do for v in base new; do
  if [ $v = base ]; then cp $base dot_amd/libdotmi.so; else cp /tmp/ab_new.so dot_amd/libdotmi.so; fi
  python bench.py --no-cpu-baseline --extra-workloads none --steps 40 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['value'], d['step_breakdown_ms'], 'bs us', 1e3*r['avg_launch_ms'], 'frac', r['frac'])"
done; done
cp /tmp/ab_new.so dot_amd/libdotmi.so
