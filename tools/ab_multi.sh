# compare several builds of libdotmi.so: bash tools/ab_multi.sh "<so list>" [bench args]
sos=$1; shift
cp dot_amd/libdotmi.so /tmp/ab_keep.so
for so in $sos; do
  cp $so dot_amd/libdotmi.so
  python bench.py --no-cpu-baseline --extra-workloads none "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$so', d['value'], d['step_breakdown_ms'], 'bs us', round(1e3*r['avg_launch_ms'],2), 'frac', r['frac'])"
done
cp /tmp/ab_keep.so dot_amd/libdotmi.so
