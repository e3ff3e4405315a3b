# bench.py on bar17K with the synchronous / asynchronous end-of-step refresh, same box
for v in 1 0 1 0; do
BENCH_SYNC_REFRESH=$v python bench.py --workload bar17K_twist --steps 20 --no-cpu-baseline --extra-workloads none 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sync_refresh=$v', d['value'], d.get('ms_per_step_p50'), d.get('step_breakdown_ms'), d['roofline']['avg_launch_ms'])"
done
