# in-situ back-solve launch time (bench.py roofline) of the q-based order, the early order and its ablations, same box
for cfg in "DOTMI_EARLY_BACKSOLVE=0" "DOTMI_EARLY_BACKSOLVE=1" "DOTMI_EARLY_ABORT=0" "DOTMI_EARLY_HOST_CTL=0" "DOTMI_EARLY_BACKSOLVE=0" "DOTMI_EARLY_BACKSOLVE=1" "DOTMI_EARLY_ABORT=0" "DOTMI_EARLY_HOST_CTL=0"; do
env $cfg python bench.py --workload bar17K_twist --steps 20 --no-cpu-baseline --extra-workloads none 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg', d['value'], r['avg_launch_ms'], r['frac'], r['launches_timed'], r.get('launches_stopped'))"
done
